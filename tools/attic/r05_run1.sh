set -x
timeout 600 python tools/split_check.py bf16x3 130 > gpurun_out/split_check.txt 2>&1; echo rc=$?
tail -5 gpurun_out/split_check.txt
timeout 300 python tools/kernel_time.py bf16x3 512 10 5 "" fold_fc=false > gpurun_out/kt_bf16x3.json 2> gpurun_out/kt_bf16x3.err; echo rc=$?
timeout 300 python tools/kernel_time.py fp32 512 10 5 "" fold_fc=false > gpurun_out/kt_fp32.json 2> gpurun_out/kt_fp32.err; echo rc=$?
cat gpurun_out/kt_bf16x3.json gpurun_out/kt_fp32.json

# (a) the shipped cell kernels in the configuration that broke the persistent experiment, (b) the reduced reproducer
mkdir -p gpurun_out
: > gpurun_out/soak_shared.jsonl
for cfg in "fp32 512 2 3e7" "fp32 512 4 3e7" "bf16_all 4096 2 1.2e8" "bf16_all 4096 4 1.2e8" "bf16_all 512 4 3e7" "bf16x3 512 4 2e7"; do
  timeout 400 python tools/soak_shared.py $cfg 300 >> gpurun_out/soak_shared.jsonl 2>> gpurun_out/soak_shared.err; echo "rc=$? $cfg"
done
cat gpurun_out/soak_shared.jsonl
timeout 900 build/ldsdma_repro 20000 40 > gpurun_out/ldsdma_repro.txt 2>&1; echo "repro rc=$?"
cat gpurun_out/ldsdma_repro.txt

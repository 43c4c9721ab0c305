DS_HIP_LIBRARY=$PWD/build/variants/lib_clk.so python3 tools/kernel_time.py bf16x3 512 1 1 lstm 2>/dev/null | grep CELL | tail -12

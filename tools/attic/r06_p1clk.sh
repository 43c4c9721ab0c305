#!/bin/bash
# round 6: where a P1 step of the split chain goes (DS_P1_CLOCK builds: s_memtime per half-step and wave), per timing-only bisect build
# usage (one gpurun call): bash tools/attic/r06_p1clk.sh <variant names: build/variants/lib_<name>.so ...>
mkdir -p gpurun_out/r06
for v in "$@"; do
  echo "== $v"
  DS_HIP_LIBRARY=$PWD/build/variants/lib_$v.so python3 tools/kernel_time.py bf16x3 512 1 1 inception 2>/dev/null | grep '^P1CLK block 0 .* W 45' | awk '{k=$5; if (seen[k] < 1) print; seen[k]++}' | sort -k5,5n | cut -c1-260 | tee gpurun_out/r06/p1clk_$v.txt
done

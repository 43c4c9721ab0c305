#!/bin/bash
# round 6: same-box A/B of split-chain variants: stand-alone chain time, then the pipelined step (three-step and folded)
# usage: bash tools/attic/r06_chain_ab.sh <variant names>
bash tools/ab.sh "bf16x3 512 10 5 inception" "$@"
for n in "$@"; do
  export DS_HIP_LIBRARY=$PWD/build/variants/lib_$n.so
  a=$(python3 tools/step_time.py bf16x3 512 100 5 fold_fc=false 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['sites_per_s_median'])")
  b=$(python3 tools/step_time.py bf16x3 512 100 5 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['sites_per_s_median'])")
  echo "$n step three-step $a folded $b"
done

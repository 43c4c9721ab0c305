#!/bin/bash
for r in 1 2; do
for n in base new; do
  if [ $n = base ]; then export DS_HIP_LIBRARY=$PWD/build/variants/lib_base.so; else unset DS_HIP_LIBRARY; fi
  for cfg in "bf16_all 4096" "bf16 4096" "bf16_all 512 400"; do
    echo -n "$n $cfg: "; python3 tools/step_time.py $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k in ('sites_per_s','median','ms_per_step','value')} or d)"
  done
done; done

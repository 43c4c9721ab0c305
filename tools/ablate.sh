#!/bin/bash
# diagnostic: rebuild the library with DS_EXP=n (timing-only ablations; results are wrong) and time the kernels
for e in "$@"; do
  make -B -C deepsignal_amd/csrc EXP=$e > /dev/null 2>&1
  echo "EXP=$e"
  DS_SERIAL=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['ms_per_step'], {k.split('<')[1] if '<' in k else k:(v['us_per_step']) for k,v in r['kernels'].items() if 'gemm' in k or 'fused' in k})
"
done
make -B -C deepsignal_amd/csrc EXP=0 > /dev/null 2>&1

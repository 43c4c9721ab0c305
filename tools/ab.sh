#!/bin/bash
# same-box A/B of kernel variants: tools/ab.sh "<kernel_time args>" name1 name2 ...   (libraries from tools/build_variant.sh)
args=$1; shift
mkdir -p gpurun_out/ab
for round in 1 2; do
for n in "$@"; do
  DS_HIP_LIBRARY=$PWD/build/variants/lib_$n.so python3 tools/kernel_time.py $args > gpurun_out/ab/$n.$round.json 2> gpurun_out/ab/$n.err
  python3 - "$n" "$round" <<'PY'
import json,sys
n,r=sys.argv[1:3]
try:
    d=json.load(open('gpurun_out/ab/%s.%s.json'%(n,r)))
    print(n,r,' '.join('%s=%.1f/%.1f'%(k.split('<')[0][:22],v['median_us_per_step'],v['min_us_per_step']) for k,v in d['kernels'].items()))
except Exception as e: print(n,r,'FAILED',e)
PY
done; done

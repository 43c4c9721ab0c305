#!/bin/bash
# same-box A/B of the round-4 library (3209d53, build/r04/) against HEAD's on the native fp32 headline.
# usage (one gpurun call): bash tools/attic/r06_item3.sh
out=gpurun_out/r06_item3; mkdir -p $out
FAST="--no-split --no-configs2 --no-cpu-baseline --no-host-path --no-profile-pass --no-standalone-pass --no-fast-mode"
for rep in 1 2 3; do
  for lib in r04 head; do
    if [ $lib = r04 ]; then export DS_HIP_LIBRARY=$PWD/build/r04/libdeepsignal_hip_r04.so; else unset DS_HIP_LIBRARY; fi
    python3 bench.py --gpus 1 --steps 20 --warmup 5 $FAST > $out/${lib}_20_$rep.json 2> $out/${lib}_20_$rep.err
  done
done
for lib in r04 head; do
  if [ $lib = r04 ]; then export DS_HIP_LIBRARY=$PWD/build/r04/libdeepsignal_hip_r04.so; else unset DS_HIP_LIBRARY; fi
  python3 bench.py --gpus 1 --steps 400 --warmup 20 $FAST > $out/${lib}_400.json 2> $out/${lib}_400.err
done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r06_item3/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d['value'], d['ms_per_step'])
    except Exception as e: print(os.path.basename(f),'FAILED',e)
PY

for v in "$@"; do
  DS_HIP_LIBRARY=$PWD/build/variants/lib_$v.so python tools/kernel_time.py fp32 512 10 5 "inception" fold_fc=false 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$v', {k[:28]: (v['median_us_per_step'], v['min_us_per_step']) for k,v in d['kernels'].items()})"
  DS_HIP_LIBRARY=$PWD/build/variants/lib_$v.so python bench.py --no-cpu-baseline --no-configs2 --no-host-path --no-fast-mode --no-split --no-profile-pass --windows 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$v bench', d['value'], d['windows']['sites_per_s'])"
done

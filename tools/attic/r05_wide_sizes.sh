# dense_split at larger forwards: the 256 x 192 tile (K in 2 / 1 ranges by max_batch) against the 128 x 96 tile (DS_TUNE_SPLIT_DENSE_NARROW),
# stand-alone us per forward. (The comparison against always four ranges -- 8 - 12 % slower from 1,024 sites -- was run with a
# temporary switch that is not in the library any more.)
for n in 1024 2048 4096; do for narrow in false true; do
echo -n "n=$n narrow=$narrow: "
python3 tools/kernel_time.py bf16x3 $n 6 3 "" fold_fc=false split_dense_narrow=$narrow 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print(' '.join('%s=%.1f'%(k.split('<')[0][:24],v['median_us_per_step']) for k,v in d['kernels'].items() if 'dense' in k or 'head_k' in k))"
done; done

# dense_split: 256 x 192 tile (DS_SPLIT_DENSE_WIDE=2: every size) against the shipped choice, stand-alone us per forward
for n in 1024 2048 4096; do for w in 1 2; do
echo -n "n=$n wide=$w: "; DS_SPLIT_DENSE_WIDE=$w python3 tools/kernel_time.py bf16x3 $n 6 3 "" fold_fc=false 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print(' '.join('%s=%.1f'%(k.split('<')[0][:24],v['median_us_per_step']) for k,v in d['kernels'].items() if 'dense' in k or 'head_k' in k))"
done; done

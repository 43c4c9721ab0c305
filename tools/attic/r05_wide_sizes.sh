# dense_split 256 x 192: K ranges by forward size (4 / 2 / 1) against always 4 (a site's bits then do not depend on the forward's size)
for n in 1024 2048 4096; do for f in 0 1; do
echo -n "n=$n fixed4=$f: "; if [ $f = 1 ]; then export DS_SPLIT_DENSE_FIXED_PARTS=1; else unset DS_SPLIT_DENSE_FIXED_PARTS; fi
python3 tools/kernel_time.py bf16x3 $n 6 3 "" fold_fc=false 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print(' '.join('%s=%.1f'%(k.split('<')[0][:24],v['median_us_per_step']) for k,v in d['kernels'].items() if 'dense' in k or 'head_k' in k))"
done; done

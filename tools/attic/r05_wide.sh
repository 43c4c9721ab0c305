for round in 1 2; do for w in 0 1; do
echo "wide tile=$w"; python3 tools/kernel_time.py bf16x3 512 10 5 "" fold_fc=false split_dense_narrow=$([ $w = 0 ] && echo true || echo false) 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print(' '.join('%s=%.1f'%(k.split('<')[0][:24],v['median_us_per_step']) for k,v in d['kernels'].items() if 'dense' in k or 'head' in k))"
done; done
python -m pytest tests/test_gpu_split.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3

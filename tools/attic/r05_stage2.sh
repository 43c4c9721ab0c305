python tools/split_check.py bf16x3 130 2>&1 | grep -E "==|lstm|fc1|logits" | head -30
for t in narrow lds1 wide; do python tools/kernel_time.py bf16x3 512 10 5 "" fold_fc=false lstm_tiling=$t 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$t', {k.split('(')[0][:34]: v['median_us_per_step'] for k,v in d['kernels'].items()})"; done
python -m pytest tests/test_gpu_split.py -x -q 2>&1 | tail -5

for n in 512 1024 4096; do for t in narrow lds1 wide; do python tools/kernel_time.py bf16x3 $n 8 3 "lstm" lstm_tiling=$t 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$n $t', {k.split('(')[0][:30]: v['median_us_per_step'] for k,v in d['kernels'].items()})"; done; done
python -m pytest tests/test_gpu_split.py -x -q 2>&1 | tail -3

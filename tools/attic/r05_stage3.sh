python tools/split_check.py bf16x3 130 2>&1 | grep -E "==|stem_conv" | head -12
python tools/kernel_time.py bf16x3 512 10 5 "" fold_fc=false 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('bf16x3', {k.split('(')[0][:30]: v['median_us_per_step'] for k,v in d['kernels'].items()})"
python tools/kernel_time.py fp32 512 10 5 "" fold_fc=false 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('fp32', {k.split('(')[0][:30]: v['median_us_per_step'] for k,v in d['kernels'].items()})"
for t in wide xwide; do python tools/kernel_time.py bf16_all 4096 6 3 "lstm" lstm_tiling=$t 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('bf16_all 4096 $t', {k.split('(')[0][:30]: v['median_us_per_step'] for k,v in d['kernels'].items()})"; done
python -m pytest tests/test_gpu_split.py tests/test_gpu_bf16.py -x -q -k "split or tile_shapes" 2>&1 | tail -3

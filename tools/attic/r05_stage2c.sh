# split LSTM tile and split dense against the fp32 kernels at 4096 and 1024 sites per forward
for n in 4096 1024; do
for t in narrow wide; do python tools/kernel_time.py bf16x3 $n 6 3 "" fold_fc=false lstm_tiling=$t split_dense_min_n=1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$n bf16x3 $t', {k.split('(')[0][:30]: v['median_us_per_step'] for k,v in d['kernels'].items() if 'lstm' in k or 'dense' in k or 'gemm' in k or 'fused' in k})"; done
python tools/kernel_time.py fp32 $n 6 3 "" fold_fc=false 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$n fp32', {k.split('(')[0][:30]: v['median_us_per_step'] for k,v in d['kernels'].items() if 'lstm' in k or 'dense' in k or 'gemm' in k or 'fused' in k})"
done

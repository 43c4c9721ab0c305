for v in "$@"; do for n in 512 2048; do DS_HIP_LIBRARY=$PWD/build/variants/lib_$v.so python tools/kernel_time.py bf16x3 $n 8 3 "dense" fold_fc=false split_dense_min_n=1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$v $n', {k.split('(')[0][:30]: (v['median_us_per_step'], v['min_us_per_step']) for k,v in d['kernels'].items()})"; done; done
for n in 512 2048; do python tools/kernel_time.py fp32 $n 8 3 "gemm" fold_fc=false 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('fp32 $n', {k.split('(')[0][:30]: (v['median_us_per_step'], v['min_us_per_step']) for k,v in d['kernels'].items()})"; done
DS_HIP_LIBRARY=$PWD/build/variants/lib_d4x1.so python -m pytest tests/test_gpu_split.py -x -q 2>&1 | tail -2

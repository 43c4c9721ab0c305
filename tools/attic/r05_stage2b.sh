for v in p0 p1 p1k2; do for t in narrow lds1 wide; do DS_HIP_LIBRARY=$PWD/build/variants/lib_$v.so python tools/kernel_time.py bf16x3 512 10 5 "_split" fold_fc=false lstm_tiling=$t 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$v $t', {k.split('(')[0][:30]: v['median_us_per_step'] for k,v in d['kernels'].items() if 'fused' not in k})"; done; done
DS_HIP_LIBRARY=$PWD/build/variants/lib_p1.so python -m pytest tests/test_gpu_split.py -x -q 2>&1 | tail -3

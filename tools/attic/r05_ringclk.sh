for v in "$@"; do echo "== $v"; DS_HIP_LIBRARY=$PWD/build/variants/lib_$v.so python3 tools/kernel_time.py bf16x3 512 2 1 dense fold_fc=false 2>/dev/null | grep RING | tail -4; done

# timing-only builds of the split chain's P1 (DS_SPLIT_BISECT bits: 1 no transform / LDS writes, 2 no row loads, 4 no barrier, 8 no weight loads, 16 no fragment reads)
for v in "$@"; do echo -n "== $v  "; DS_HIP_LIBRARY=$PWD/build/variants/lib_$v.so python3 tools/stamps.py bf16x3 512 2>/dev/null | grep -A2 "module 5" | grep "wave 0" | cut -c1-110; done

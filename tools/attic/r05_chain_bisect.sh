for v in "$@"; do echo "== $v"; DS_HIP_LIBRARY=$PWD/build/variants/lib_$v.so python3 tools/stamps.py bf16x3 512 2>/dev/null | grep -A2 "module 5"; done
bash tools/ab.sh "bf16x3 512 10 3 inception" "$@" 2>&1 | grep -v -e Warning -e amdgpu.ids

# P1 bisect builds of the split kernel: stamps of module 2 per variant (timing only; results of bisect builds are wrong)
for v in "$@"; do
  echo "== $v"
  DS_HIP_LIBRARY=$PWD/build/variants/lib_$v.so timeout 120 python tools/stamps.py bf16x3 512 2>/dev/null | sed -n 2,3p
done

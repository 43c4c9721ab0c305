#!/bin/bash
# Build a kernel variant of the library for A/B timing: tools/build_variant.sh <name> [-DMACRO=VALUE ...]
#   -> build/variants/lib_<name>.so   (run with DS_HIP_LIBRARY=<that path> python tools/kernel_time.py ...)
# DS_VARIANT_TU names the translation unit the macros apply to (default ds_kernels.hip; ds_split.hip for the split-operand
# kernels); the other translation units are compiled once into build/obj and re-used while their sources do not change.
set -e
name=$1; shift
tu=${DS_VARIANT_TU:-ds_kernels.hip}
mkdir -p build/variants build/obj
cd deepsignal_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-result -Wno-pass-failed"
objs=""
for src in ds_kernels.hip ds_split.hip ds_engine.cpp ds_io.cpp; do
  extra=""
  [ "$src" = "ds_split.hip" ] && extra="-fno-slp-vectorize"      # as the Makefile builds it
  if [ "$src" = "$tu" ]; then
    o=../../build/obj/${src%.*}.$name.o
    /opt/rocm/bin/hipcc $FLAGS $extra "$@" -c -x hip $src -o $o
  else
    o=../../build/obj/${src%.*}.o
    if [ ! -f $o ] || [ -n "$(find $src ds_internal.h ds_device.h ds_dec_token.h ../../include/deepsignal_hip.h -newer $o)" ]; then
      /opt/rocm/bin/hipcc $FLAGS $extra -c -x hip $src -o $o
    fi
  fi
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $objs -o ../../build/variants/lib_$name.so

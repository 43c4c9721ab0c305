#!/bin/bash
# Build a kernel variant of the library for A/B timing: tools/build_variant.sh <name> [-DMACRO=VALUE ...]
#   -> build/variants/lib_<name>.so   (run with DS_HIP_LIBRARY=<that path> python tools/kernel_time.py ...)
set -e
name=$1; shift
mkdir -p build/variants
cd deepsignal_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-result "$@" -shared -x hip ds_kernels.hip ds_engine.cpp ds_io.cpp -o ../../build/variants/lib_$name.so

#!/bin/bash
# Variant of the library that differs ONLY in objnerf_train32.o (the fp32 fused kernel), beside the product one:
#   tools/build_t32_variant.sh abl8 -DOBJ32_ABL=8     ->  openobj_amd/csrc/variants/libobjnerf_hip_t32_abl8.so
# (13 s per variant instead of the whole library; use with OBJNERF_LIB=<that path>.)  T32_SCHED overrides the unit's
# scheduling flags.
set -e
name=$1; shift
cd "$(dirname "$0")/../openobj_amd/csrc"
make -s
mkdir -p variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS ${T32_SCHED--mllvm -amdgpu-sched-strategy=iterative-ilp -fno-honor-nans} "$@" -c objnerf_train32.hip -o variants/t32_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libobjnerf_hip_t32_$name.so $(ls *.o | grep -v '^objnerf_train32\.o$') variants/t32_$name.o
echo built openobj_amd/csrc/variants/libobjnerf_hip_t32_$name.so

#!/bin/bash
# Diagnostic: rebuild ONLY the second-generation bf16 kernel with extra flags and link it with the product's other objects:
#   tools/build_v2_variant.sh name [flags...]  ->  openobj_amd/csrc/variants/libobjnerf_hip_name.so  (use with OBJNERF_LIB)
# SCHED (environment) overrides the scheduling-strategy flags of that unit (default: the Makefile's).
set -e
name=$1; shift
cd "$(dirname "$0")/../openobj_amd/csrc"
mkdir -p variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS ${SCHED--mllvm -amdgpu-sched-strategy=iterative-ilp} "$@" -c objnerf_train_bf16v2.hip -o variants/v2_$name.o
others=$(ls *.o | grep -v objnerf_train_bf16v2.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libobjnerf_hip_$name.so variants/v2_$name.o $others
echo built variants/libobjnerf_hip_$name.so

#!/bin/bash
# Diagnostic: rebuild ONLY the second-generation bf16 kernel with extra flags and link it with the product's other objects:
#   tools/build_v2_variant.sh name [flags...]  ->  openobj_amd/csrc/variants/libobjnerf_hip_name.so  (use with OBJNERF_LIB)
# SCHED (environment) overrides the scheduling-strategy flags of that unit (default: the Makefile's).
set -e
name=$1; shift
cd "$(dirname "$0")/../openobj_amd/csrc"
mkdir -p variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
unit=objnerf_train_bf16v2${V2F:+f}      # V2F=1: the feature-loss instantiation instead
/opt/rocm/bin/hipcc $FLAGS ${SCHED--mllvm -amdgpu-sched-strategy=iterative-ilp -fno-honor-nans} "$@" -c $unit.hip -o variants/v2_$name.o
others=$(ls *.o | grep -v "^$unit.o\$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libobjnerf_hip_$name.so variants/v2_$name.o $others
echo built variants/libobjnerf_hip_$name.so

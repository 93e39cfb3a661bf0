#!/bin/bash
# Diagnostic builds of the library with extra -D flags on the fused-kernel translation units:
#   tools/build_variant.sh NAME -DPHASE_TIMING ...   ->  openobj_amd/csrc/abl/lib_NAME.so   (use with OBJNERF_LIB=...)
set -e
cd "$(dirname "$0")/../openobj_amd/csrc"
name=$1; shift
mkdir -p abl
make -s
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off"
SCHED32=${SCHED32--mllvm -amdgpu-sched-strategy=iterative-ilp}
/opt/rocm/bin/hipcc $FLAGS -mllvm -amdgpu-sched-strategy=max-ilp "$@" -c objnerf_train.hip -o abl/train_$name.o &
/opt/rocm/bin/hipcc $FLAGS $SCHED32 "$@" -c objnerf_train32.hip -o abl/train32_$name.o &
/opt/rocm/bin/hipcc $FLAGS "$@" -c objnerf_train_bf16.hip -o abl/train_bf16_$name.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/lib_$name.so abl/train_$name.o abl/train32_$name.o abl/train_bf16_$name.o objnerf_misc.o objnerf_generic.o objnerf_helpers.o
echo built abl/lib_$name.so

#!/bin/bash
# Diagnostic build of the library with extra -D flags, beside the product one:
#   tools/build_variant.sh pe63 -DOBJ_PE_ANCHORS=63   ->  openobj_amd/csrc/variants/libobjnerf_hip_pe63.so
# Use it with OBJNERF_LIB=<that path> (openobj_amd/_lib.py).  Never loaded by default.
set -e
name=$1; shift
cd "$(dirname "$0")/../openobj_amd/csrc"
out=variants/$name
mkdir -p $out
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
pids=()
for f in objnerf_train objnerf_train32 objnerf_train_bf16 objnerf_train_bf16v2 objnerf_train_bf16v2f objnerf_misc objnerf_generic objnerf_helpers objnerf_train256 objnerf_render objnerf_render_bf16; do
  extra=""
  [ $f = objnerf_train ] && extra="-mllvm -amdgpu-sched-strategy=max-ilp"
  [ $f = objnerf_train32 ] && extra="${T32_SCHED--mllvm -amdgpu-sched-strategy=iterative-ilp -fno-honor-nans}"   # T32_SCHED: override (may be empty)
  [ $f = objnerf_train_bf16 ] && extra="${T16_SCHED--mllvm -amdgpu-sched-strategy=iterative-ilp}"   # T16_SCHED: override for the bf16 fused kernels
  [ $f = objnerf_train_bf16v2 ] && extra="${T16V2_SCHED--mllvm -amdgpu-sched-strategy=iterative-ilp -fno-honor-nans}"   # T16V2_SCHED: the second-generation bf16 kernel
  [ $f = objnerf_train_bf16v2f ] && extra="${T16V2F_SCHED--mllvm -amdgpu-sched-strategy=iterative-ilp -fno-honor-nans}"
  [ $f = objnerf_generic ] && extra="$TGEN_SCHED"   # TGEN_SCHED: flags for the layer-wise / small-batch unit only
  [ $f = objnerf_train256 ] && extra="-mllvm -amdgpu-mfma-vgpr-form -Wno-inline-asm $T256_EXTRA"   # T256_EXTRA: flags for that unit only
  /opt/rocm/bin/hipcc $FLAGS $extra "$@" -c $f.hip -o $out/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libobjnerf_hip_$name.so $out/*.o
echo built openobj_amd/csrc/variants/libobjnerf_hip_$name.so

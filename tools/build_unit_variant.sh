#!/bin/bash
# Variant of the library that differs in ONE translation unit, beside the product one (seconds instead of the whole build):
#   tools/build_unit_variant.sh objnerf_train post62 -DFEAT_POST_PF=6 -DFEAT_POST_PFF=2 -DFEAT_POST_WPE=2
#     -> openobj_amd/csrc/variants/libobjnerf_hip_objnerf_train_post62.so      (use with OBJNERF_LIB=<that path>)
# The unit keeps the Makefile's own extra flags (scheduling strategy etc.).
set -e
unit=$1; name=$2; shift 2
cd "$(dirname "$0")/../openobj_amd/csrc"
make -s
mkdir -p variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
case $unit in
  objnerf_train) X="-mllvm -amdgpu-sched-strategy=max-ilp";;
  objnerf_train32|objnerf_train_bf16v2|objnerf_train_bf16v2f) X="-mllvm -amdgpu-sched-strategy=iterative-ilp -fno-honor-nans";;
  objnerf_train_bf16) X="-mllvm -amdgpu-sched-strategy=iterative-ilp";;
  objnerf_train256) X="-mllvm -amdgpu-mfma-vgpr-form -Wno-inline-asm";;
  objnerf_train256r) X="-Wno-inline-asm";;
  *) X="";;
esac
/opt/rocm/bin/hipcc $FLAGS $X "$@" -c $unit.hip -o variants/${unit}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libobjnerf_hip_${unit}_$name.so $(ls *.o | grep -v "^$unit\.o$") variants/${unit}_$name.o
echo built openobj_amd/csrc/variants/libobjnerf_hip_${unit}_$name.so

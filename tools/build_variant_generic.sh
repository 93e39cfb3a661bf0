#!/bin/bash
# Diagnostic build with extra -D flags on objnerf_generic.hip:  tools/build_variant_generic.sh NAME -DOBJ_GEMM_BK=32
set -e
cd "$(dirname "$0")/../openobj_amd/csrc"
name=$1; shift
mkdir -p abl
make -s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off "$@" -c objnerf_generic.hip -o abl/generic_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/lib_$name.so objnerf_train.o objnerf_train32.o objnerf_train_bf16.o objnerf_misc.o objnerf_helpers.o abl/generic_$name.o
echo built abl/lib_$name.so

#!/bin/bash
# Diagnostic builds of the one-launch small-batch kernel (objnerf_small_body.h, SM_ABL bits) -> openobj_amd/csrc/abl/lib_sm<bits>.so
# then (on the GPU box):  for v in ...; OBJNERF_LIB=.../lib_sm$v.so python3 tools/bg_trace.py --metric [--bf16]
set -e
cd "$(dirname "$0")/../openobj_amd/csrc"
mkdir -p abl
make -s
OBJS=$(ls *.o | grep -v '^objnerf_generic\.o$' | tr '\n' ' ')      # every unit of the Makefile but the one rebuilt below
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -DSM_ABL=$v -c objnerf_generic.hip -o abl/generic_sm$v.o &
done
wait
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/lib_sm$v.so $OBJS abl/generic_sm$v.o
done
ls -la abl/lib_sm*.so

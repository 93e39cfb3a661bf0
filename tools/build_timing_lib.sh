#!/bin/bash
# diagnostic build: libobjnerf_hip_timing.so = the library with objnerf_train256.hip compiled -DOBJ256_TIMING (kernel A
# prints its phase ticks); use with OBJNERF_LIB=$PWD/openobj_amd/csrc/libobjnerf_hip_timing.so
set -e
cd "$(dirname "$0")/../openobj_amd/csrc"
make -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -Wno-inline-asm -DOBJ256_TIMING $EXTRA -c objnerf_train256.hip -o /tmp/t256_timing.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function -Wno-inline-asm -DOBJ256_TIMING $EXTRA -c objnerf_train256r.hip -o /tmp/t256r_timing.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libobjnerf_hip_timing.so $(ls *.o | grep -v "objnerf_train256.o\|objnerf_train256r.o") /tmp/t256_timing.o /tmp/t256r_timing.o
ls -la libobjnerf_hip.so libobjnerf_hip_timing.so

#!/bin/bash
# Run ON THE GPU BOX: time bench.py (bf16, objects only) for each variant library given by name; prints ms per step / kernel ms.
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=${BENCH_ARGS:---dtype bf16 --no-bg --steps 40 --warmup 10 --no-psnr --no-cpu-baseline --no-other-configs --no-peak --no-bf16-line}
for n in "$@"; do
  lib=$R/openobj_amd/csrc/variants/libobjnerf_hip_$n.so
  [ "$n" = product ] && lib=$R/openobj_amd/csrc/libobjnerf_hip.so
  OBJNERF_LIB=$lib python3 $R/bench.py $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"
done

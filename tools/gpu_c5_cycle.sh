#!/bin/bash
# c5 iteration cycle on the GPU: phase ticks, spec check, bench line with the feature loss
OBJNERF_LIB=$PWD/openobj_amd/csrc/libobjnerf_hip_timing.so timeout 300 python tools/c5_timing.py fp16 > gpurun_out/c5_timing.txt 2>&1
grep -E "^---|t256" gpurun_out/c5_timing.txt | tail -4
timeout 700 python tools/c5feat_check.py ${CHECK_ARGS---full} > gpurun_out/r05_c5feat_check.txt 2>&1; grep -E "worst|Error|error" gpurun_out/r05_c5feat_check.txt | tr '\n' ' '; echo
for f in "" "--feat"; do
timeout 600 python bench.py --config c5 $f --dtype fp16 --steps 5 --warmup 2 --no-cpu-baseline --no-psnr --no-other-configs --no-peak 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('c5 $f', j[\"value\"], j[\"ms_per_step\"])"
done

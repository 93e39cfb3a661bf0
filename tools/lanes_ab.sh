#!/bin/bash
# same-box comparison of the chunk-stream arrangements of ops.TrainWorkspace (configs[4], fp16)
run() { timeout 600 python bench.py --config c5 --dtype fp16 --steps 8 --warmup 3 --no-cpu-baseline --no-psnr --no-other-configs --no-peak 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$1', round(j['ms_per_step'],2), j['config']['chunk_streams'])"; }
for i in 1 2; do
OBJNERF_ONE_LANE=1 run one_lane
run two_lanes
done

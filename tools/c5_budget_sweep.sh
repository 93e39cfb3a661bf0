#!/bin/bash
# configs[4] share (64 objects, hidden 256, 8192 x 128, fp16) against the workspace budget = objects per chunk (two lanes while
# a second buffer leaves a quarter of the device free): gpurun_out/r06_c5/budget_sweep.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_c5; mkdir -p $O; cd $R
Q="--no-cpu-baseline --no-psnr --no-peak --no-other-configs --no-bf16-line --no-bg"
for gib in 48 64 80 96 120 160; do
  for feat in "" "--feat"; do
    OBJNERF_WORKSPACE_BUDGET_GIB=$gib timeout 600 python3 bench.py --config c5 --dtype fp16 $feat --steps 4 --warmup 2 $Q --detail-out $O/d.json 2>$O/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=json.load(open('$O/d.json'))['config']
print('budget %3d GiB %-6s chunks %d lanes %d  step_ms %.1f' % ($gib, '$feat', c['object_chunks'], c['chunk_streams'], d['ms_per_step']))" || tail -2 $O/err.txt
  done
done | tee $O/budget_sweep.txt

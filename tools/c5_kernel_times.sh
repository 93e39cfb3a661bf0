#!/bin/bash
# GPU box: per-kernel averages of the configs[4] share (8 objects) under rocprofv3 --stats.  usage: tools/c5_kernel_times.sh [dtype] [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}
DT=${1:-fp16}; TAG=${2:-c5}
OUT=$R/gpurun_out/c5k_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python3 $R/bench.py --config c5 --dtype $DT --objects 8 --no-bg --no-bf16-line --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-psnr --no-peak --no-other-configs > $OUT/bench.json 2> $OUT/err.txt
f=$(ls $OUT/*kernel_stats.csv $OUT/*/*kernel_stats.csv 2>/dev/null | head -1)
head -8 "$f" | cut -c1-170
rm -f $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv

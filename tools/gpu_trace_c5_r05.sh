#!/bin/bash
# Run ON THE GPU BOX (gpurun): kernel stats of c5 (fp16) without and with the feature loss
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace5c5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-psnr --no-peak --no-other-configs"
run() {   # name, bench args
  local n=$1; shift 1
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$n -o s -- python3 $R/bench.py "$@" $Q > $OUT/bench_$n.json 2> $OUT/bench_$n.err
  cp $(ls $OUT/t_$n/*kernel_stats.csv | head -1) $OUT/kernel_stats_$n.csv
  rm -rf $OUT/t_$n
  head -14 $OUT/kernel_stats_$n.csv | cut -c1-200
}
run c5_fp16 --config c5 --dtype fp16 --steps 5 --warmup 2
run c5feat_fp16 --config c5 --feat --dtype fp16 --steps 5 --warmup 2

#!/bin/bash
# Run ON THE GPU BOX (gpurun): kernel trace of the default bench (fp32 then bf16 steps) and of the c4 share, with one
# step's timeline printed (tools/kernel_timeline.py) and the per-kernel stats kept.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-psnr --no-peak --no-other-configs"
run() {   # name, anchor, bench args
  local n=$1 a=$2; shift 2
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$n -o s -- python3 $R/bench.py "$@" $Q > $OUT/bench_$n.json 2> $OUT/bench_$n.err
  python3 $R/tools/kernel_timeline.py $OUT/t_$n "$a" 40 > $OUT/timeline_$n.txt 2>&1
  cp $(ls $OUT/t_$n/*kernel_stats.csv | head -1) $OUT/kernel_stats_$n.csv
  rm -rf $OUT/t_$n
}
run default_f32 train_fused32 --steps 20 --warmup 5 --no-bf16-line
run default_bf16 train_fused_bf16v2 --steps 20 --warmup 5 --dtype bf16
run c4share_bf16 train_fused_bf16v2f --config c4 --objects 15 --bg-ranks 8 --steps 20 --warmup 5 --dtype bf16
timeout 300 python3 $R/bench.py --steps 100 --warmup 20 $Q > $OUT/bench_default_plain.json 2>/dev/null
timeout 300 python3 $R/bench.py --config c4 --objects 15 --bg-ranks 8 --steps 100 --warmup 20 $Q > $OUT/bench_c4share_plain.json 2>/dev/null
timeout 300 python3 $R/tools/mapping_bench.py > $OUT/mapping_bench.txt 2>&1
ls -la $OUT

#!/bin/bash
# Run ON THE GPU BOX (gpurun): the rocprofv3 evidence of round 6 for profiles/ (kernel stats + PMC passes).
# Counters are collected in their own passes with --kernel-trace only; every profiler run is under `timeout`.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-psnr --no-peak --no-other-configs"
T="timeout 600"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"
SQ2="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM"
stats() {  # name, bench args
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$1 -o s -- python3 $R/bench.py ${@:2} $Q > $OUT/bench_$1_under_rocprof.json 2> $OUT/bench_$1.err
}
pmc() {    # name, kernel substring, bench args: four passes (FETCH_SIZE, WRITE_SIZE, SQ1, SQ2)
  local n=$1 k=$2; shift 2
  for pass in fetch write sq sq2; do
    case $pass in fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; sq) C="$SQ1";; sq2) C="$SQ2";; esac
    $T rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${n}_$pass -o p -- python3 $R/bench.py "$@" --no-bf16-line --steps 3 --warmup 1 $Q > /dev/null 2> $OUT/pmc_${n}_$pass.err
    f=$(ls $OUT/pmc_${n}_$pass/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f "$k" > $OUT/pmc_${n}_$pass.txt
    rm -f $OUT/pmc_${n}_$pass/*.csv
  done
}
# ONLY=c4full (environment): just the PMC passes of the full configs[3] launch (120 objects), merged into an existing prof5/
if [ "${ONLY:-all}" = c4full ]; then
  pmc c4full_f32 "train_fused32_kernel<true" --config c4 --no-bg
  pmc c4full_bf16 train_fused_bf16v2f --config c4 --no-bg --dtype bf16
  ls -la $OUT | tail -12
  exit 0
fi
stats default --gpus 1 --steps 20 --warmup 5
stats feat --config c3 --steps 10 --warmup 3
stats c4share --config c4 --objects 15 --bg-ranks 8 --steps 10 --warmup 3
stats c5 --config c5 --dtype fp16 --objects 8 --no-bg --no-bf16-line --steps 2 --warmup 1
pmc f32 train_fused32 --no-bg
pmc bf16 train_fused_bf16 --no-bg --dtype bf16
pmc c3_f32 "train_fused32_kernel<true" --config c3 --no-bg
pmc c3_bf16 train_fused_bf16v2f --config c3 --no-bg --dtype bf16
pmc c4_f32 "train_fused32_kernel<true" --config c4 --objects 15 --bg-ranks 8 --no-bg
pmc c4_bf16 train_fused_bf16v2f --config c4 --objects 15 --bg-ranks 8 --no-bg --dtype bf16
# the one-launch background iteration (round 6): its three kernels alone (tools/bg_trace.py at the benchmark's 1200 x 64 samples)
pmc_bg() {   # name, kernel substring, bg_trace args
  local n=$1 k=$2; shift 2
  for pass in fetch write sq sq2; do
    case $pass in fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; sq) C="$SQ1";; sq2) C="$SQ2";; esac
    STEPS=5 $T rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${n}_$pass -o p -- python3 $R/tools/bg_trace.py "$@" > /dev/null 2> $OUT/pmc_${n}_$pass.err
    f=$(ls $OUT/pmc_${n}_$pass/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f "$k" > $OUT/pmc_${n}_$pass.txt
    rm -f $OUT/pmc_${n}_$pass/*.csv
  done
}
pmc_bg bgsmall_f32 "train_small_kernel<4, false, false>" --metric
pmc_bg bgsmall_bf16 "train_small_kernel<4, true, false>" --metric --bf16
pmc_bg bggroup_f32 "gemm_group_kernel" --metric
for v in "" "--bf16" "--feat" "--feat --bf16"; do
  tag=$(echo "bg$v" | tr -d ' -')
  STEPS=50 $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$tag -o s -- python3 $R/tools/bg_trace.py --metric $v > $OUT/bgtrace_$tag.txt 2> /dev/null
  tagn=$(echo "bgnative$v" | tr -d ' -')
  STEPS=50 $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$tagn -o s -- python3 $R/tools/bg_trace.py $v > $OUT/bgtrace_$tagn.txt 2> /dev/null
done
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_mapping -o s -- python3 $R/tools/mapping_bench.py > $OUT/mapping_bench.txt 2> /dev/null
timeout 600 python3 $R/bench.py --steps 100 --warmup 20 $Q > $OUT/bench_default_plain.json 2>/dev/null
timeout 300 python3 $R/bench.py --steps 100 --warmup 20 --no-pipeline $Q > $OUT/bench_default_nopipe.json 2>/dev/null
timeout 300 python3 $R/bench.py --config c4 --objects 15 --bg-ranks 8 --steps 100 --warmup 20 $Q > $OUT/bench_c4share.json 2>/dev/null
for kk in fwd256_kernel wgrad256_kernel; do :; done
C5="--config c5 --dtype fp16 --objects 8 --no-bg"
for pass in fetch write sq sq2; do
  case $pass in fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; sq) C="$SQ1";; sq2) C="$SQ2";; esac
  $T rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_c5_$pass -o p -- python3 $R/bench.py $C5 --no-bf16-line --steps 2 --warmup 1 $Q > /dev/null 2> $OUT/pmc_c5_$pass.err
  f=$(ls $OUT/pmc_c5_$pass/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    python3 $R/tools/pmc_summary.py $f fwd256_kernel > $OUT/pmc_c5_${pass}_fwd256.txt
    python3 $R/tools/pmc_summary.py $f wgrad256_kernel > $OUT/pmc_c5_${pass}_wgrad256.txt
  fi
  rm -f $OUT/pmc_c5_$pass/*.csv
done
timeout 300 python3 $R/bench.py --no-bg --steps 100 --warmup 20 $Q > $OUT/bench_nobg.json 2>/dev/null
timeout 300 python3 $R/bench.py --config c3 --steps 30 --warmup 5 $Q > $OUT/bench_c3.json 2>/dev/null
timeout 300 python3 $R/bench.py --config c4 --steps 20 --warmup 3 $Q > $OUT/bench_c4.json 2>/dev/null
timeout 600 python3 $R/bench.py --config c5 --dtype fp16 --steps 5 --warmup 1 $Q > $OUT/bench_c5_fp16.json 2>/dev/null
timeout 600 python3 $R/bench.py --config c5 --dtype bf16 --steps 5 --warmup 1 $Q > $OUT/bench_c5_bf16.json 2>/dev/null
OBJNERF_DIST_SELFTEST=1 timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 1 --steps 20 --warmup 5 $Q 2>/dev/null | tail -1 > $OUT/bench_dist_selftest.json
rm -f $OUT/stats_*/*kernel_trace.csv
ls -la $OUT

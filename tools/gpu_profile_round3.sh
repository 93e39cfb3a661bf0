#!/bin/bash
# Run ON THE GPU BOX (gpurun): the rocprofv3 evidence of round 3 for profiles/ (kernel stats + PMC passes).
# Counters are collected in their own passes with --kernel-trace only; every profiler run is under `timeout`.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-psnr --no-peak --no-other-configs"
P="--no-bg --no-bf16-line --steps 3 --warmup 1 $Q"
C5="--config c5 --dtype fp16 --objects 8 --no-bg --no-bf16-line --steps 2 --warmup 1 $Q"
T="timeout 600"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -o s -- python3 $R/bench.py --steps 10 --warmup 3 $Q > $OUT/bench_default_under_rocprof.json 2> $OUT/bench_default.err
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_feat -o s -- python3 $R/bench.py --config c3 --steps 10 --warmup 3 $Q > $OUT/bench_feat_under_rocprof.json 2> $OUT/bench_feat.err
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -o s -- python3 $R/bench.py $C5 > $OUT/bench_c5_under_rocprof.json 2> $OUT/bench_c5.err
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $R/bench.py $P > /dev/null 2> $OUT/pmc_fetch.err
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python3 $R/bench.py $P > /dev/null 2> $OUT/pmc_write.err
$T rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -o p -- python3 $R/bench.py $P > /dev/null 2> $OUT/pmc_sq.err
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o p -- python3 $R/bench.py $P > /dev/null 2> $OUT/pmc_sq2.err
$T rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS --output-format csv -d $OUT/pmc_bf16 -o p -- python3 $R/bench.py --dtype bf16 $P > /dev/null 2> $OUT/pmc_bf16.err
# configs[4] share (8 objects = one workspace chunk): HBM bytes and issue counters of the two hidden-256 kernels
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_c5_fetch -o p -- python3 $R/bench.py $C5 > /dev/null 2> $OUT/pmc_c5_fetch.err
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_c5_write -o p -- python3 $R/bench.py $C5 > /dev/null 2> $OUT/pmc_c5_write.err
$T rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS --output-format csv -d $OUT/pmc_c5_sq -o p -- python3 $R/bench.py $C5 > /dev/null 2> $OUT/pmc_c5_sq.err
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_c5_sq2 -o p -- python3 $R/bench.py $C5 > /dev/null 2> $OUT/pmc_c5_sq2.err
timeout 300 python3 $R/bench.py --no-bg --steps 10 --warmup 3 $Q > $OUT/bench_nobg.json 2>/dev/null
timeout 300 python3 $R/bench.py --config c3 --steps 10 --warmup 3 $Q > $OUT/bench_c3.json 2>/dev/null
timeout 300 python3 $R/bench.py --config c4 --steps 5 --warmup 2 $Q > $OUT/bench_c4.json 2>/dev/null
timeout 600 python3 $R/bench.py --config c5 --steps 2 --warmup 1 $Q --no-bf16-line > $OUT/bench_c5_f32.json 2>/dev/null
timeout 600 python3 $R/bench.py --config c5 --dtype fp16 --steps 5 --warmup 1 $Q > $OUT/bench_c5_fp16.json 2>/dev/null
timeout 600 python3 $R/bench.py --config c5 --dtype bf16 --steps 5 --warmup 1 $Q > $OUT/bench_c5_bf16.json 2>/dev/null
OBJNERF_DIST_SELFTEST=1 timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 1 --steps 10 --warmup 3 $Q 2>/dev/null | tail -1 > $OUT/bench_dist_selftest.json
for d in pmc_fetch pmc_write pmc_sq pmc_sq2; do
  f=$(ls $OUT/$d/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f train_fused32 > $OUT/$d.txt
done
f=$(ls $OUT/pmc_bf16/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 $R/tools/pmc_summary.py $f train_fused_bf16 > $OUT/pmc_bf16.txt
for d in pmc_c5_fetch pmc_c5_write pmc_c5_sq pmc_c5_sq2; do
  f=$(ls $OUT/$d/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    python3 $R/tools/pmc_summary.py $f fwd256_kernel > $OUT/${d}_fwd256.txt
    python3 $R/tools/pmc_summary.py $f wgrad256_kernel > $OUT/${d}_wgrad256.txt
  fi
done
rm -f $OUT/*/*kernel_trace.csv $OUT/pmc_*/*counter_collection.csv   # large; the summaries above are what is kept
ls -la $OUT

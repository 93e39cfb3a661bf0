#!/bin/bash
# Run ON THE GPU BOX (gpurun): refresh the rocprofv3 evidence for profiles/ (kernel stats + PMC passes).
# Counters are collected in their own passes with --kernel-trace only (never with sys/hip trace domains).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-psnr --no-peak"
P="--no-bg --no-bf16-line --steps 3 --warmup 1 $Q"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -o s -- python3 $R/bench.py --steps 10 --warmup 3 $Q > $OUT/bench_default.json 2> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_feat -o s -- python3 $R/bench.py --config c3 --steps 10 --warmup 3 $Q > $OUT/bench_feat.json 2> $OUT/bench_feat.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $R/bench.py $P > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python3 $R/bench.py $P > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -o p -- python3 $R/bench.py $P > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o p -- python3 $R/bench.py $P > /dev/null 2> $OUT/pmc_sq2.err
# the bf16-operand kernel: instruction counts for its VALU roofline (bench.py bf16_mode.roofline)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS --output-format csv -d $OUT/pmc_bf16 -o p -- python3 $R/bench.py --dtype bf16 $P > /dev/null 2> $OUT/pmc_bf16.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_bf16b -o p -- python3 $R/bench.py --dtype bf16 $P > /dev/null 2> $OUT/pmc_bf16b.err
python3 $R/bench.py --no-bg --steps 10 --warmup 3 $Q > $OUT/bench_nobg.json 2>/dev/null
python3 $R/bench.py > $OUT/bench_full.json 2>$OUT/bench_full.err
python3 $R/bench.py --config c3 --steps 10 --warmup 3 $Q > $OUT/bench_c3.json 2>/dev/null
python3 $R/bench.py --config c4 --steps 5 --warmup 2 $Q > $OUT/bench_c4.json 2>/dev/null
python3 $R/bench.py --config c5 --steps 2 --warmup 1 $Q --no-bf16-line > $OUT/bench_c5.json 2>/dev/null
python3 $R/bench.py --config c5 --dtype fp16 --steps 2 --warmup 1 $Q > $OUT/bench_c5_fp16.json 2>/dev/null
python3 $R/bench.py --config c5 --dtype bf16 --steps 2 --warmup 1 $Q > $OUT/bench_c5_bf16.json 2>/dev/null
# the layer-wise kernels of the configs[4] share in bf16 mode (resident-panel GEMMs, split-K weight gradients)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5_bf16 -o s -- python3 $R/bench.py --config c5 --dtype bf16 --objects 8 --steps 2 --warmup 1 --no-bg $Q > /dev/null 2> $OUT/stats_c5_bf16.err
# fp32 layer chain against its three-bf16-piece emulation (tools/ubench_x3.hip, built in-tree before the call)
[ -x $R/openobj_amd/csrc/abl/ubench_x3 ] && $R/openobj_amd/csrc/abl/ubench_x3 > $OUT/ubench_x3.txt 2>&1
# the two-collective iteration over RCCL with one rank (the multi-GPU code path on the one GPU there is)
OBJNERF_DIST_SELFTEST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 1 --steps 10 --warmup 3 $Q 2>/dev/null | tail -1 > $OUT/bench_dist_selftest.json
for d in pmc_fetch pmc_write pmc_sq pmc_sq2; do
  f=$(ls $OUT/$d/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f train_fused32 > $OUT/$d.txt
done
for d in pmc_bf16 pmc_bf16b; do
  f=$(ls $OUT/$d/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f train_fused_bf16 > $OUT/$d.txt
done
rm -f $OUT/*/*kernel_trace.csv $OUT/pmc_*/*counter_collection.csv   # large; the summaries above are what is kept
ls -la $OUT

#!/bin/bash
# round 6, session 4: the feature-loss chain after the folds (gram + snapshot in feat_pre_kernel, head finish in finalize_kernel,
# O column on the VALU): parity tests of the feature paths, the object chain alone (bench --no-bg: kernel_ms = HIP events around
# objnerf_train_step) for the default library and the feat_post variants, one traced step of c3 / c4 share.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s4; mkdir -p $O
cd $R
python3 -m pytest tests/test_hip_parity.py tests/test_round5_gpu.py tests/test_16bit_spec_gpu.py tests/test_api_gpu.py -m gpu -x -q 2>&1 | tail -15 > $O/feat_tests.txt; tail -4 $O/feat_tests.txt
Q="--no-cpu-baseline --no-psnr --no-peak --no-other-configs --no-bf16-line"
chain() {  # label, lib
  local label=$1 lib=$2
  for cfg in "c3 f32" "c3 bf16" "c4s f32" "c4s bf16" "c2 bf16"; do
    set -- $cfg
    case $1 in c3) A="--config c3";; c4s) A="--config c4 --objects 15 --bg-ranks 8";; c2) A="";; esac
    if [ -n "$lib" ]; then export OBJNERF_LIB=$lib; else unset OBJNERF_LIB; fi
    python3 bench.py $A --dtype $2 --no-bg --steps 30 --warmup 5 $Q --detail-out $O/d.json 2>$O/chain_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-8s %-4s %-5s chain_ms %.3f step_ms %.3f' % ('$label', '$1', '$2', d['roofline']['kernel_ms'], d['ms_per_step']))"
  done
}
{
chain default ""
chain post622 $R/openobj_amd/csrc/variants/libobjnerf_hip_objnerf_train_post622.so
chain post424 $R/openobj_amd/csrc/variants/libobjnerf_hip_objnerf_train_post424.so
} | tee $O/feat_chain.txt
unset OBJNERF_LIB
cd /tmp && export TMPDIR=/tmp
run() {   # name, anchor, bench args
  local n=$1 a=$2; shift 2
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$n -o s -- python3 $R/bench.py "$@" $Q > $O/bench_$n.json 2> $O/bench_$n.err
  python3 $R/tools/kernel_timeline.py $O/t_$n "$a" 40 > $O/timeline_$n.txt 2>&1
  cp $(ls $O/t_$n/*kernel_stats.csv | head -1) $O/kernel_stats_$n.csv
  rm -rf $O/t_$n
}
run c3_bf16 train_fused_bf16v2f --config c3 --steps 10 --warmup 3 --dtype bf16
run c3_f32 "train_fused32_kernel<true" --config c3 --steps 10 --warmup 3
run c4share_bf16 train_fused_bf16v2f --config c4 --objects 15 --bg-ranks 8 --steps 10 --warmup 3 --dtype bf16
run default_bf16 train_fused_bf16v2_kernel --steps 10 --warmup 3 --dtype bf16
cat $O/timeline_c3_bf16.txt | head -24
for v in "" "--bf16" "--feat" "--feat --bf16"; do echo "bg chain alone (tools/bg_trace.py --metric $v): $(STEPS=100 python3 $R/tools/bg_trace.py --metric $v 2>/dev/null | tail -1)"; done | tee $O/bg_chain.txt
python3 - <<'P'
import csv,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r06_s4/kernel_stats_*.csv')):
    print(os.path.basename(f))
    for r in list(csv.DictReader(open(f)))[:14]:
        print('  %-70s calls %5s avg_us %9.1f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
P

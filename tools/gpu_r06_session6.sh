#!/bin/bash
# round 6, session 6: the background chain's launch folds (grouped preparation GEMMs + FeatPrepTask, T / M as one group, the
# head's finish + AdamW in one kernel): parity tests of the feature / background paths, launches per step from kernel stats,
# the chains' times.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_s6; mkdir -p $O; cd $R
python3 -m pytest tests/test_hip_parity.py tests/test_round5_gpu.py tests/test_16bit_spec_gpu.py tests/test_api_gpu.py tests/test_bf16_gpu.py -m gpu -x -q 2>&1 | tail -8 > $O/tests.txt; tail -4 $O/tests.txt
for v in "--feat" "--feat --bf16"; do echo "bg chain alone (tools/bg_trace.py --metric $v): $(STEPS=100 python3 tools/bg_trace.py --metric $v 2>/dev/null | tail -1)"; done | tee $O/bg_chain.txt
for v in "--feat" "--feat --bf16"; do echo "bg chain alone, native shape (tools/bg_trace.py $v): $(STEPS=200 python3 tools/bg_trace.py $v 2>/dev/null | tail -1)"; done | tee -a $O/bg_chain.txt
Q="--no-cpu-baseline --no-psnr --no-peak --no-other-configs --no-bf16-line"
cd /tmp && export TMPDIR=/tmp
run() {   # name, anchor, bench args
  local n=$1 a=$2; shift 2
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$n -o s -- python3 $R/bench.py "$@" $Q > $O/bench_$n.json 2> $O/bench_$n.err
  python3 $R/tools/kernel_timeline.py $O/t_$n "$a" 40 > $O/timeline_$n.txt 2>&1
  cp $(ls $O/t_$n/*kernel_stats.csv | head -1) $O/kernel_stats_$n.csv
  rm -rf $O/t_$n
}
run c3_bf16 train_fused_bf16v2f --config c3 --steps 10 --warmup 3 --dtype bf16
run c4share_bf16 train_fused_bf16v2f --config c4 --objects 15 --bg-ranks 8 --steps 10 --warmup 3 --dtype bf16
python3 - <<'P'
import csv,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r06_s6/kernel_stats_*.csv')):
    rows=list(csv.DictReader(open(f)))
    steps=[int(r['Calls']) for r in rows if 'train_fused' in r['Name']][0]
    print(os.path.basename(f), 'steps', steps)
    tot=0
    for r in rows:
        c=int(r['Calls'])
        if c>=steps:
            tot+=c/steps
            print('  %-70s per step %4.1f avg_us %9.1f' % (r['Name'][:70], c/steps, float(r['AverageNs'])/1e3))
    print('  launches per step (kernels with >= 1 call per step): %.1f' % tot)
P
cd $R
for cfg in "--config c3 --dtype bf16" "--config c4 --objects 15 --bg-ranks 8 --dtype bf16" "--config c3"; do
  python3 bench.py $cfg --steps 30 --warmup 5 $Q --detail-out $O/d.json 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$cfg: step_ms %.3f chain_ms %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done | tee $O/steps.txt

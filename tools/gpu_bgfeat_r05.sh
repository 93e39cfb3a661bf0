#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for m in "" "--bf16"; do for s in "--metric" ""; do echo "bg_trace --feat $s $m: $(STEPS=200 python3 tools/bg_trace.py --feat $s $m | tail -1)"; done; done 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bgs -o s -- python3 $R/tools/bg_trace.py --metric --feat > /dev/null 2>&1
python3 - <<'P'
import csv,glob
f=glob.glob('/tmp/bgs/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1))
P

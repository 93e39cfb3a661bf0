cd $GRAFT_REPO_ROOT
for v in base 1 2 4 8 16 32 63; do
  if [ $v = base ]; then L=""; else L="OBJNERF_LIB=$GRAFT_REPO_ROOT/openobj_amd/csrc/abl/lib_sm$v.so"; fi
  a=$(env $L STEPS=200 python3 tools/bg_trace.py --metric | tail -1)
  b=$(env $L STEPS=200 python3 tools/bg_trace.py --metric --bf16 | tail -1)
  c=$(env $L STEPS=200 python3 tools/bg_trace.py | tail -1)
  echo "abl $v | metric f32: $a | metric bf16: $b | native f32: $c"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bgs -o s -- python3 $GRAFT_REPO_ROOT/tools/bg_trace.py --metric > /dev/null 2>&1
python3 - <<'P'
import csv,glob
f=glob.glob('/tmp/bgs/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]: print(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3)
P

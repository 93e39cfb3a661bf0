# Diagnostic (GPU box): kernel sequence with durations of the last launches of a bench.py run -> gpurun_out/seq/seq.txt
#   bash tools/kernel_seq.sh --config c5 --dtype bf16 --objects 4 --steps 1 --warmup 1
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/seq; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o s -- python3 $R/bench.py "$@" --no-psnr --no-cpu-baseline --no-peak > $OUT/bench.json 2> $OUT/err.txt
python3 - <<'PY'
import csv, glob, os
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
f=glob.glob(R+'/gpurun_out/seq/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
out=open(R+'/gpurun_out/seq/seq.txt','w')
for r in rows[-160:]:
    n=r['Kernel_Name']
    n=n[:40]+'..'+n[-46:] if len(n)>90 else n
    out.write("%-90s grid %s,%s,%s start %10.1f us  dur %9.1f us\n"%(n, r['Grid_Size_X'],r['Grid_Size_Y'],r['Grid_Size_Z'],(int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
rm -f $OUT/*kernel_trace.csv

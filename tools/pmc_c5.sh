# Diagnostic (GPU box): PMC passes over the layer-wise kernels of the configs[4] share in bf16 mode -> gpurun_out/c5pmc/summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/c5pmc; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --config c5 --dtype bf16 --objects 2 --steps 1 --warmup 1 --no-psnr --no-cpu-baseline --no-peak --no-bg"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o p -- python3 $B > /dev/null 2> $OUT/f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o p -- python3 $B > /dev/null 2> $OUT/w.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/s1 -o p -- python3 $B > /dev/null 2> $OUT/s1.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/s2 -o p -- python3 $B > /dev/null 2> $OUT/s2.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $OUT/t -o p -- python3 $B > /dev/null 2> $OUT/t.err
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); OUT=R+'/gpurun_out/c5pmc'
out=open(OUT+'/summary.txt','w')
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('f','w','s1','s2','t'):
    fs=glob.glob(OUT+'/'+d+'/*counter_collection.csv')
    if not fs: out.write(d+': no file\n'); continue
    for r in csv.DictReader(open(fs[0])):
        n=r['Kernel_Name']
        if 'gemm_bf16' not in n: continue
        key=n[n.index('gemm_bf16_kernel')+16:][:52]+' grid '+r['Grid_Size']
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()):
    out.write(k+'\n')
    for c,vals in sorted(v.items()): out.write('   %-28s n=%3d mean=%.5g\n'%(c,len(vals),sum(vals)/len(vals)))
PY
rm -rf $OUT/f $OUT/w $OUT/s1 $OUT/s2 $OUT/t

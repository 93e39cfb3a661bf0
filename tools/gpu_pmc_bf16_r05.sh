#!/bin/bash
# Run ON THE GPU BOX: SQ counter passes of the bf16 second-generation kernels (separate --pmc passes, kernel trace only)
# + their kernel times.  usage: gpu_pmc_bf16_r05.sh TAG
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-a}
OUT=$R/gpurun_out/pmc5_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-psnr --no-peak --no-other-configs"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"
SQ2="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM"
pmc() {
  local n=$1 k=$2; shift 2
  for pass in sq sq2; do
    case $pass in sq) C="$SQ1";; sq2) C="$SQ2";; esac
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${n}_$pass -o p -- python3 $R/bench.py "$@" --no-bf16-line --steps 3 --warmup 1 $Q > /dev/null 2> $OUT/pmc_${n}_$pass.err
    f=$(ls $OUT/pmc_${n}_$pass/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f "$k" > $OUT/pmc_${n}_$pass.txt
    rm -rf $OUT/pmc_${n}_$pass
  done
}
pmc bf16 train_fused_bf16v2_kernel --no-bg --dtype bf16
pmc c3_bf16 train_fused_bf16v2f --config c3 --no-bg --dtype bf16
timeout 300 python3 $R/bench.py --no-bg --dtype bf16 --steps 100 --warmup 20 --no-bf16-line $Q > $OUT/bench_nobg_bf16.json 2>/dev/null
timeout 300 python3 $R/bench.py --config c3 --no-bg --dtype bf16 --steps 50 --warmup 10 --no-bf16-line $Q > $OUT/bench_c3_nobg_bf16.json 2>/dev/null
cat $OUT/pmc_*_sq2.txt
python3 - <<P
import json
for n in ['nobg_bf16','c3_nobg_bf16']:
    d=json.loads(open('$OUT/bench_%s.json'%n).read().strip().split('\n')[-1]); print(n, '%.2f M rays/s %.3f ms kernel %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']))
P

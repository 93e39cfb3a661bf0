#!/bin/bash
# Run ON THE GPU BOX (gpurun): PMC passes (separate runs, --kernel-trace only beside --pmc) of configs[4]'s kernels --
# the row-split kernel A (fwdr256_kernel), the first form (OBJ256_FIRST_FORM=1) and kernel B -- 8 objects per launch.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc5c5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T="timeout 600"
Q="--no-cpu-baseline --no-psnr --no-peak --no-other-configs"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"
SQ2="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM"
C5="--config c5 --dtype fp16 --objects 8 --no-bg"
: > $OUT/pmc_c5.txt
for form in rowsplit firstform; do
  [ $form = firstform ] && export OBJ256_FIRST_FORM=1
  kern=fwdr256_kernel; [ $form = firstform ] && kern=fwd256_kernel
  for pass in fetch write sq sq2; do
    case $pass in fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; sq) C="$SQ1";; sq2) C="$SQ2";; esac
    $T rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p_${form}_$pass -o p -- python3 $R/bench.py $C5 --no-bf16-line --steps 2 --warmup 1 $Q > /dev/null 2> $OUT/p_${form}_$pass.err
    f=$(ls $OUT/p_${form}_$pass/*counter_collection.csv 2>/dev/null | head -1)
    if [ -n "$f" ]; then
      echo "# $form, pass $pass, $kern" >> $OUT/pmc_c5.txt; python3 $R/tools/pmc_summary.py $f $kern >> $OUT/pmc_c5.txt
      if [ $form = rowsplit ]; then echo "# pass $pass, wgrad256_kernel" >> $OUT/pmc_c5.txt; python3 $R/tools/pmc_summary.py $f wgrad256_kernel >> $OUT/pmc_c5.txt; fi
    fi
    rm -rf $OUT/p_${form}_$pass
  done
done
cat $OUT/pmc_c5.txt

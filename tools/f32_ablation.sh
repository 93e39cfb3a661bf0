#!/bin/bash
# Ceiling measurements of the fp32 headline kernel (train_fused32_kernel<false, false, 64>, c2: 50 x 4096 x 64) on the GPU
# box: every openobj_amd/csrc/variants/libobjnerf_hip_t32_abl*.so (tools/build_t32_variant.sh ablN -DOBJ32_ABL=N, bits in
# objnerf_train32.hip) timed by bench.py's own HIP events around objnerf_train_step, no background network, and the
# PHASE_TIMING build's per-phase ticks.  -> gpurun_out/r06_f32/ablation.txt
O=gpurun_out/r06_f32; mkdir -p $O
B="--steps 30 --warmup 5 --no-bg --no-cpu-baseline --no-bf16-line --no-psnr --no-peak --no-other-configs"
run() {  # name, lib ("" = product)
  if [ -n "$2" ]; then export OBJNERF_LIB=$2; else unset OBJNERF_LIB; fi
  python3 bench.py $B --detail-out $O/detail_$1.json 2> $O/err_$1.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-10s kernel_ms %7.3f  step_ms %7.3f  status %s' % ('$1', r['kernel_ms'], d['ms_per_step'], d['config']['loss_status']))"
}
{
run base ""
for so in $(ls openobj_amd/csrc/variants/libobjnerf_hip_t32_*.so | sort -V); do
  n=$(basename $so .so); n=${n#libobjnerf_hip_t32_}
  [ $n = PHASE ] && continue
  run $n $PWD/$so
done
} | tee $O/ablation.txt
if [ -f openobj_amd/csrc/variants/libobjnerf_hip_t32_PHASE.so ]; then
  OBJNERF_LIB=$PWD/openobj_amd/csrc/variants/libobjnerf_hip_t32_PHASE.so python3 tools/phase_timing.py > $O/phase_ticks.txt 2>&1
  cat $O/phase_ticks.txt
fi

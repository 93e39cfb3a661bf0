#!/bin/bash
# Ceiling measurements of the bf16 headline kernel (train_fused_bf16v2_kernel, c2: 50 x 4096 x 64), as tools/f32_ablation.sh
# for the fp32 one: every openobj_amd/csrc/variants/libobjnerf_hip_v2abl*.so (tools/build_v2_variant.sh v2ablN -DV2_ABL=N; bits in
# objnerf_bf16_common.h) timed by bench.py --dtype bf16 --no-bg (HIP events around objnerf_train_step).
O=gpurun_out/r06_bf16; mkdir -p $O
B="--dtype bf16 --steps 30 --warmup 5 --no-bg --no-cpu-baseline --no-bf16-line --no-psnr --no-peak --no-other-configs"
run() {
  if [ -n "$2" ]; then export OBJNERF_LIB=$2; else unset OBJNERF_LIB; fi
  python3 bench.py $B --detail-out $O/detail_$1.json 2> $O/err_$1.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-10s kernel_ms %7.3f  step_ms %7.3f' % ('$1', r['kernel_ms'], d['ms_per_step']))"
}
{
run base ""
for so in $(ls openobj_amd/csrc/variants/libobjnerf_hip_v2abl*.so | sort -V); do
  n=$(basename $so .so); n=${n#libobjnerf_hip_}
  run $n $PWD/$so
done
} | tee $O/ablation.txt

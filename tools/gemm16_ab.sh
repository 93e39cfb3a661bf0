# Diagnostic: A/B of variant builds of the library (openobj_amd/csrc/abl/lib_G*.so, OBJNERF_LIB) on the shapes the
# 16-bit layer-wise GEMMs serve: the background step in bf16 mode (default bench) and the configs[4] share.
for so in openobj_amd/csrc/abl/lib_G*.so; do
  echo "== $so"
  OBJNERF_LIB=$PWD/$so python bench.py --dtype bf16 --steps 20 --warmup 5 --no-psnr --no-cpu-baseline --no-peak 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 c2', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  for dt in fp16 bf16; do
    OBJNERF_LIB=$PWD/$so python bench.py --config c5 --dtype $dt --objects ${C5_OBJECTS:-8} --steps 2 --warmup 1 --no-psnr --no-cpu-baseline --no-peak --no-bg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$dt c5', d['value'], d['ms_per_step'])"
  done
done

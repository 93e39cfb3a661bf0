for so in openobj_amd/csrc/abl/lib_G*.so; do
  echo "== $so"
  OBJNERF_LIB=$PWD/$so python bench.py --dtype bf16 --steps 20 --warmup 5 --no-psnr --no-cpu-baseline --no-peak 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 c2', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  OBJNERF_LIB=$PWD/$so python bench.py --config c5 --dtype fp16 --steps 2 --warmup 1 --no-psnr --no-cpu-baseline --no-peak --no-bg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp16 c5', d['value'], d['ms_per_step'])"
done

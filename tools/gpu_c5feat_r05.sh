#!/bin/bash
# c5 with the feature loss on the fused hidden-256 path: spec check at full size, the affected test files, bench lines
mkdir -p gpurun_out
timeout 900 python tools/c5feat_check.py --full > gpurun_out/r05_c5feat_check.txt 2>&1
timeout 1200 python -m pytest tests/test_fp16_gpu.py tests/test_16bit_spec_gpu.py -x -q -m gpu > gpurun_out/r05_c5feat_tests.txt 2>&1
for d in fp16 bf16; do
  timeout 600 python bench.py --config c5 --dtype $d --steps 5 --warmup 2 --no-cpu-baseline --no-psnr --no-other-configs --no-peak > gpurun_out/r05_c5_$d.json 2> gpurun_out/r05_c5_$d.err
  timeout 600 python bench.py --config c5 --feat --dtype $d --steps 5 --warmup 2 --no-cpu-baseline --no-psnr --no-other-configs --no-peak > gpurun_out/r05_c5feat_$d.json 2> gpurun_out/r05_c5feat_$d.err
done
tail -5 gpurun_out/r05_c5feat_tests.txt
for f in gpurun_out/r05_c5_fp16.json gpurun_out/r05_c5feat_fp16.json gpurun_out/r05_c5_bf16.json gpurun_out/r05_c5feat_bf16.json; do python - $f <<'P'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); print(sys.argv[1], j['value'], j['unit'], j['ms_per_step'], j.get('roofline',{}).get('frac'), j['config'])
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
tail -3 gpurun_out/r05_c5feat_fp16.err

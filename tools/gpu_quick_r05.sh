#!/bin/bash
# Run ON THE GPU BOX: the round's quick measurement set (bench lines without PSNR / CPU legs, the mapping loop, one
# background step alone).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/quick5
mkdir -p $OUT
cd $R
Q="--no-cpu-baseline --no-psnr --no-peak --no-other-configs"
for m in "" "--bf16"; do for s in "--metric" ""; do echo "bg_trace $s $m: $(STEPS=200 python3 tools/bg_trace.py $s $m | tail -1)"; done; done 2>&1 | grep -v amdgpu.ids | tee $OUT/bg_trace.txt
timeout 300 python3 bench.py --steps 100 --warmup 20 $Q > $OUT/bench_default.json 2>/dev/null
timeout 300 python3 bench.py --steps 100 --warmup 20 --no-pipeline $Q > $OUT/bench_default_nopipe.json 2>/dev/null
timeout 300 python3 bench.py --config c4 --objects 15 --bg-ranks 8 --steps 100 --warmup 20 $Q > $OUT/bench_c4share.json 2>/dev/null
timeout 300 python3 bench.py --config c3 --steps 30 --warmup 5 $Q > $OUT/bench_c3.json 2>/dev/null
timeout 300 python3 tools/mapping_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/mapping_bench.txt
python3 - <<P
import json
for n in ['default','default_nopipe','c4share','c3']:
    try:
        d=json.loads(open('$OUT/bench_%s.json'%n).read().strip().split('\n')[-1]); b=d.get('bf16_mode',{})
        print(n, 'f32 %.2f M rays/s %.3f ms (kernel %.3f) | bf16 %.2f M %.3f ms (kernel %.3f)' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], b.get('value',0)/1e6, b.get('ms_per_step',0), b.get('roofline',{}).get('kernel_ms',0)))
    except Exception as e: print(n, 'failed', e)
P
tail -3 $OUT/mapping_bench.txt

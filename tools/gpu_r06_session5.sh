R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_s5; mkdir -p $O; cd $R
python3 -m pytest tests/test_hip_parity.py tests/test_round5_gpu.py tests/test_16bit_spec_gpu.py tests/test_api_gpu.py tests/test_bf16_gpu.py tests/test_psnr_gpu.py -m gpu -x -q 2>&1 | tail -8 > $O/tests.txt; tail -5 $O/tests.txt
for v in "" "--bf16" "--feat" "--feat --bf16"; do echo "bg chain alone (tools/bg_trace.py --metric $v): $(STEPS=100 python3 tools/bg_trace.py --metric $v 2>/dev/null | tail -1)"; done | tee $O/bg_chain.txt
for v in "" "--bf16"; do echo "bg chain alone, native shape (tools/bg_trace.py $v): $(STEPS=200 python3 tools/bg_trace.py $v 2>/dev/null | tail -1)"; done | tee -a $O/bg_chain.txt
python3 bench.py --steps 20 --warmup 5 --detail-out $O/bench_detail.json > $O/bench_line.json 2> $O/bench_err.txt; python3 -c "
import json; d=json.load(open('$O/bench_detail.json')); print(json.dumps(d['summary'], indent=0))"

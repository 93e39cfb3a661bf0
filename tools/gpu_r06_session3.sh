#!/bin/bash
# round 6, session 3: the whole GPU suite on the build with finalize v2 / gram + head finish folded, the default bench, the
# remaining ablation combinations, the reproducibility report.
O=gpurun_out/r06_s3; mkdir -p $O
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/gpu_tests.txt; tail -6 $O/gpu_tests.txt
python3 tools/repro256.py > $O/repro256.txt 2>&1; tail -3 $O/repro256.txt
python3 bench.py --steps 20 --warmup 5 --detail-out $O/bench_detail.json > $O/bench_line.json 2> $O/bench_err.txt; cat $O/bench_line.json
bash tools/f32_ablation.sh > $O/ablation2.txt 2>&1; grep kernel_ms $O/ablation2.txt

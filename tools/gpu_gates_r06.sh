#!/bin/bash
# Round 6 gate records (on the GPU box): the paired hidden-256 PSNR gate re-recorded from a green run (psnr_scene.PAIRED_GATE),
# the handicap demonstration at the strengths that matter, the full-size reproducibility report.  -> gpurun_out/r06_gates/
O=gpurun_out/r06_gates; mkdir -p $O
python3 -m pytest tests/test_psnr_gpu.py -k "paired_early" -q -s 2>&1 | grep -v "^$" > $O/h256_paired_psnr.txt; tail -3 $O/h256_paired_psnr.txt
python3 tools/h256_handicap.py --modes fp16 bf16 f32 --strengths 0 0.01 0.02 0.05 0.1 0.25 > $O/h256_handicap.txt 2> $O/h256_handicap.err; cat $O/h256_handicap.txt
python3 tools/repro256.py > $O/repro256.txt 2>&1; tail -3 $O/repro256.txt
python3 -m pytest tests/test_fp16_gpu.py -k "bit_reproducible" tests/test_round5_gpu.py -q 2>&1 | tail -3

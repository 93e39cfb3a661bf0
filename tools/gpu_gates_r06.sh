#!/bin/bash
# Round 6 gate records (on the GPU box): the new full-size reproducibility test, the status word on every path, the paired
# hidden-256 PSNR gate re-recorded from a green run, and the handicap demonstration.  Outputs under gpurun_out/r06_gates/.
O=gpurun_out/r06_gates; mkdir -p $O
python3 -m pytest tests/test_fp16_gpu.py -k "bit_reproducible" tests/test_round5_gpu.py -k "status or bit_reproducible" -q -x 2>&1 | tail -5 > $O/new_tests.txt
python3 -m pytest tests/test_psnr_gpu.py -k "paired_early" -q -s 2>&1 | grep -v "^$" > $O/h256_paired_psnr.txt
python3 tools/h256_handicap.py > $O/h256_handicap.txt 2> $O/h256_handicap.err
tail -3 $O/new_tests.txt; tail -4 $O/h256_paired_psnr.txt; cat $O/h256_handicap.txt

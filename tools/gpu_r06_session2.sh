O=gpurun_out/r06_gates; mkdir -p $O
python3 tools/repro256.py > $O/repro256.txt 2>&1; tail -30 $O/repro256.txt
OBJ256_FIRST_FORM=1 python3 tools/repro256.py --feat 0 > $O/repro256_first_form.txt 2>&1; tail -12 $O/repro256_first_form.txt
python3 -m pytest tests/test_fp16_gpu.py -k "bit_reproducible" -q -x 2>&1 | tail -40 > $O/new_tests_fp16.txt; tail -40 $O/new_tests_fp16.txt
python3 -m pytest tests/test_round5_gpu.py -k "status" -q -x 2>&1 | tail -15 > $O/new_tests_status.txt; tail -5 $O/new_tests_status.txt
bash tools/f32_ablation.sh
python3 tools/h256_handicap.py --modes fp16 bf16 --strengths 0.02 0.05 0.1 > $O/h256_handicap_fine.txt 2>$O/h256_handicap_fine.err; cat $O/h256_handicap_fine.txt

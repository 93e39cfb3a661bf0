R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_s7; mkdir -p $O; cd $R
V=$R/openobj_amd/csrc/variants/libobjnerf_hip_objnerf_generic_g16wide.so
for lib in "" $V; do
  if [ -n "$lib" ]; then export OBJNERF_LIB=$lib; L=g16wide; else unset OBJNERF_LIB; L=product; fi
  for v in "--metric --bf16" "--metric --feat --bf16" "--bf16" "--feat --bf16"; do echo "$L bg chain (bg_trace.py $v): $(STEPS=200 python3 tools/bg_trace.py $v 2>/dev/null | tail -1)"; done
done | tee $O/g16_ab.txt
OBJNERF_LIB=$V python3 -m pytest tests/test_hip_parity.py tests/test_round5_gpu.py tests/test_bf16_gpu.py -m gpu -x -q 2>&1 | tail -4

#!/bin/bash
# What the weight staging of train_small_kernel costs, split: SM_ABL 16 = no staging at all, 64 = no LOADS (the LDS images
# are still written), 128 = no LDS image WRITES (the loads are still issued and waited for).  Variant libraries:
#   for v in 16 64 128; do tools/build_unit_variant.sh objnerf_generic sm$v -DSM_ABL=$v; done
# Prints the background step alone (tools/bg_trace.py) at the benchmark and the native shape, fp32 and bf16.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for L in product sm16 sm64 sm128; do
  if [ $L = product ]; then unset OBJNERF_LIB; else export OBJNERF_LIB=$R/openobj_amd/csrc/variants/libobjnerf_hip_objnerf_generic_$L.so; fi
  a=$(STEPS=200 python3 tools/bg_trace.py --metric 2>/dev/null | tail -1); b=$(STEPS=200 python3 tools/bg_trace.py --metric --bf16 2>/dev/null | tail -1)
  c=$(STEPS=300 python3 tools/bg_trace.py 2>/dev/null | tail -1); d=$(STEPS=300 python3 tools/bg_trace.py --bf16 2>/dev/null | tail -1)
  echo "$L | metric f32: $a | metric bf16: $b | native f32: $c | native bf16: $d"
done

#!/bin/bash
# kernel durations of the hidden-256 path (8 full-size objects per step): row-split kernel A vs the first form
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/c5r
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-new old}; do
  [ $v = old ] && export OBJ256_FIRST_FORM=1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$v -o s -- python3 $R/tools/c5r_check.py --time --time-only > $OUT/log_$v.txt 2>&1
  cp $(ls $OUT/t_$v/*kernel_stats.csv | head -1) $OUT/kernel_stats_$v.csv; rm -rf $OUT/t_$v
  echo "== $v"; grep "^time" $OUT/log_$v.txt; grep -E "fwd|wgrad" $OUT/kernel_stats_$v.csv | cut -c1-120
done

#!/bin/bash
for t in 16 24 32 40 48 64 96; do
  echo "== target $t MB"
  OBJ256_WG_TARGET_MB=$t timeout 300 python tools/c5r_check.py --time --time-only 2>&1 | grep "^time"
done

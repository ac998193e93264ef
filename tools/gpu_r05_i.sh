#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
touch build/variants/libinstrument.so
for CH in 0 0; do
  echo "######## LIODOM_CHAIN=$CH"
  LIODOM_DEBUG_CLOCKS=1 LIODOM_CHAIN=$CH timeout 300 python tools/gpu_debug.py clocks 2>&1 | grep -A40 "k_lm_solve (it 1) eval"
done > $OUT/r05_i_clocks.txt 2>&1
cat $OUT/r05_i_clocks.txt

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
touch build/variants/libinstrument.so
for CH in 1 0 1 0 1 0; do
  echo "######## LIODOM_CHAIN=$CH"
  LIODOM_CHAIN=$CH timeout 300 python tools/gpu_debug.py ovclocks 2>&1 | grep -v "bins\|^modes\|====" 
done > $OUT/r05_h_ovclocks.txt 2>&1
cat $OUT/r05_h_ovclocks.txt

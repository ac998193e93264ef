#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
touch build/variants/libinstrument.so
for i in 1 2; do LIODOM_CHAIN=1 timeout 300 python tools/gpu_debug.py ovclocks 2>&1 | grep -v "^modes\|====" ; done > $OUT/r05_r_ovclocks.txt 2>&1
cat $OUT/r05_r_ovclocks.txt

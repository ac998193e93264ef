#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
for HOG in spin matmul; do
  echo "##### $HOG"; ( time timeout 600 python tools/soak_two_process.py vlp16 10000 150 $HOG ) 2>&1 | grep -v "^generating\|^modes" | tail -12
done > $OUT/r05_l_soak.txt 2>&1
cat $OUT/r05_l_soak.txt
for i in 1 2 3; do LIODOM_CHAIN=1 python3 tools/replay_trace.py hdl64 300; LIODOM_CHAIN=0 python3 tools/replay_trace.py hdl64 300; done

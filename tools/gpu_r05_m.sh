#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
{
for W in ouster128 vlp16; do
  for CFG in "LIODOM_CHAIN=0" "LIODOM_CHAIN=1" "LIODOM_CHAIN=1 LIODOM_KNN_OVERLAP=2" "LIODOM_CHAIN=0 LIODOM_KNN_OVERLAP=2"; do
    for i in 1 2; do echo -n "$CFG  "; env $CFG python3 tools/replay_trace.py $W 300; done
  done
done
} > $OUT/r05_m_shapes.txt 2>&1
cat $OUT/r05_m_shapes.txt

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for sh in hdl64 vlp16 ouster128; do timeout 600 python tools/overlap_equal.py $sh ${EQ_SCANS:-300} 2>&1 | tail -6; done > $OUT/r05_e2_equal.txt 2>&1
cat $OUT/r05_e2_equal.txt
AB_WORKLOADS="hdl64 vlp16 ouster128" AB_ROUNDS=3 tools/gpu_ab.sh r05_e2 "" base product:LIODOM_SPECULATE=0 product product:LIODOM_SPECULATE=2

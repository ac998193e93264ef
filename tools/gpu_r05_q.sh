#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
{
timeout 300 python tools/overlap_equal.py hdl64 300 2>&1 | tail -1
timeout 300 python tools/overlap_equal.py ouster128 120 2>&1 | tail -1
timeout 900 python tools/ab_value.py --rounds 4 --workload hdl64 --scans 240 product:LIODOM_CHAIN=1 product:LIODOM_CHAIN=0
timeout 600 python tools/ab_value.py --rounds 2 --workload ouster128 --scans 240 product:LIODOM_CHAIN=1 product:LIODOM_CHAIN=0
} > $OUT/r05_q_ab.txt 2>&1
cat $OUT/r05_q_ab.txt

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
{
timeout 600 python tools/ab_value.py --rounds 5 --workload hdl64 --scans 240 product:LIODOM_CHAIN=1 product:LIODOM_CHAIN=0
timeout 600 python tools/ab_value.py --rounds 3 --workload hdl64 --scans 20 product:LIODOM_CHAIN=1 product:LIODOM_CHAIN=0
timeout 600 python tools/ab_value.py --rounds 3 --workload vlp16 --scans 240 product:LIODOM_CHAIN=1 product:LIODOM_CHAIN=0
} > $OUT/r05_o_ab.txt 2>&1
cat $OUT/r05_o_ab.txt

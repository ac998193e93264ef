#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for CH in 1 0; do
  rm -rf $OUT/prof_g
  LIODOM_CHAIN=$CH timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_g -- python3 $R/tools/replay_trace.py hdl64 150 > $OUT/r05_g_trace$CH.log 2>&1
  grep "scans/s" $OUT/r05_g_trace$CH.log
  DB=$(find $OUT/prof_g -name "*.db" | head -1)
  python3 $R/tools/chain_phases.py $DB 100 > $OUT/r05_g_phases$CH.txt 2>&1
  python3 $R/tools/timeline2.py $DB 60 > $OUT/r05_g_timeline$CH.txt 2>&1
  cat $OUT/r05_g_phases$CH.txt
  rm -rf $OUT/prof_g
done
for CH in 1 0; do LIODOM_CHAIN=$CH python3 $R/tools/replay_trace.py hdl64 300; done

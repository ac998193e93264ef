#!/bin/bash
# chain mode: equality, rates, trace
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 300 python tools/overlap_equal.py hdl64 300 > $OUT/r05_f_equal.txt 2>&1; tail -5 $OUT/r05_f_equal.txt
for CH in 1 0; do
  echo "=== LIODOM_CHAIN=$CH"; LIODOM_CHAIN=$CH timeout 300 python tools/dropin_rates.py hdl64 200 5 2>&1 | tail -4
done > $OUT/r05_f_rates.txt 2>&1
cat $OUT/r05_f_rates.txt
cd /tmp
rm -rf $OUT/prof_f
LIODOM_CHAIN=1 timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_f -- python3 $R/tools/workload_run.py hdl64 1 80 > $OUT/r05_f_trace.log 2>&1
DB=$(find $OUT/prof_f -name "*.db" | head -1)
python3 $R/tools/timeline2.py $DB 70 > $OUT/r05_f_timeline.txt 2>&1
rm -rf $OUT/prof_f
head -70 $OUT/r05_f_timeline.txt

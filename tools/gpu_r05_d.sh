#!/bin/bash
# timeline of the host-fed replay with event-ordered (0) and flag-ordered (1) uploads
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for M in 0 1; do
  rm -rf $OUT/prof_hf$M
  LIODOM_UPLOAD_FLAGS=$M timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/prof_hf$M -- python3 $R/tools/hostfed_run.py 80 > $OUT/r05_d_hf$M.log 2>&1
  tail -2 $OUT/r05_d_hf$M.log
  DB=$(find $OUT/prof_hf$M -name "*.db" | head -1)
  python3 $R/tools/timeline2.py $DB 90 > $OUT/r05_d_timeline_hf$M.txt 2>&1
  rm -rf $OUT/prof_hf$M
done

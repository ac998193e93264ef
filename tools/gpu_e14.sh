#!/bin/bash
# kernel durations of the four-launch chain (no overlap): HEAD~2 library vs the product
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for lib in base product; do
  rm -rf $OUT/prof_e14
  cat > /tmp/run_e14.py <<PY
import os, sys
sys.path.insert(0, "$R")
import liodom_amd.api as api
if "$lib" != "product":
    api._LIB = os.path.join("$R", "build", "variants", "lib$lib.so"); api.is_stale = lambda: False
sys.argv = ["replay_trace.py", "hdl64", "150"]
exec(open("$R/tools/replay_trace.py").read())
PY
  LIODOM_KNN_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_e14 -- python3 /tmp/run_e14.py > $OUT/r05_e14_run_$lib.txt 2>&1
  DB=$(find $OUT/prof_e14 -name "*.db" | head -1)
  { echo "######## $lib"; tail -1 $OUT/r05_e14_run_$lib.txt; python3 $R/tools/rocprof_summary.py $DB | head -14; } > $OUT/r05_e14_$lib.txt 2>&1
  cat $OUT/r05_e14_$lib.txt
done
rm -rf $OUT/prof_e14

#!/bin/bash
# round-5 first visit: any-order probe, GPU suite with durations, baseline bench + kernel trace
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
nproc > $OUT/r05_a_nproc.txt; env | grep -i "omp\|thread" >> $OUT/r05_a_nproc.txt
timeout 120 build/tools/anyorder_probe > $OUT/r05_a_anyorder.txt 2>&1
cat $OUT/r05_a_anyorder.txt
( time timeout 1500 python -m pytest tests -m gpu -q --durations=30 ) > $OUT/r05_a_pytest_durations.log 2>&1
tail -45 $OUT/r05_a_pytest_durations.log
tools/gpu_step.sh r05_a none_xyz > $OUT/r05_a_step.log 2>&1
tail -30 $OUT/r05_a_step.log

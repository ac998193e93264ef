#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
{
for S in 0 1 2; do timeout 60 build/tools/queue_probe 6 $S; done
GPU_MAX_HW_QUEUES=8 timeout 60 build/tools/queue_probe 6 0
GPU_MAX_HW_QUEUES=2 timeout 60 build/tools/queue_probe 6 0
} > $OUT/r05_c_queue_probe.txt 2>&1
cat $OUT/r05_c_queue_probe.txt

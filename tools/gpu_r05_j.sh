#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
run() { echo "=== $*"; env "$@" timeout 300 python tools/dropin_rates.py hdl64 200 3 2>&1 | grep -v "^hdl64" | tail -3; }
{
run LIODOM_UPLOAD_FLAGS=1 LIODOM_HOSTFED_OVERLAP=1 LIODOM_CHAIN=1
run LIODOM_UPLOAD_FLAGS=1 LIODOM_HOSTFED_OVERLAP=1 LIODOM_CHAIN=0
run LIODOM_UPLOAD_FLAGS=1 LIODOM_HOSTFED_OVERLAP=0 LIODOM_CHAIN=1
run LIODOM_UPLOAD_FLAGS=0 LIODOM_CHAIN=1 DROPIN_HOSTMALLOC=1
} > $OUT/r05_j_dropin.txt 2>&1
cat $OUT/r05_j_dropin.txt

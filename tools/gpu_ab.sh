#!/bin/bash
# One GPU-box visit for an A/B: quick parity tests, then tools/ab_value.py over the arms given.  usage: tools/gpu_ab.sh <tag> "<pytest -k expr>" arm arm ...
set -u
TAG=$1; KEXPR=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
if [ -n "$KEXPR" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $OUT/${TAG}_pytest.log 2>&1; tail -4 $OUT/${TAG}_pytest.log; fi
for wl in ${AB_WORKLOADS:-hdl64}; do
  timeout 900 python tools/ab_value.py --rounds ${AB_ROUNDS:-3} --workload $wl "$@" > $OUT/${TAG}_ab_$wl.txt 2>&1
  tail -8 $OUT/${TAG}_ab_$wl.txt
done

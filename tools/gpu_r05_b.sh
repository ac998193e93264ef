#!/bin/bash
# round-5 visit b: flag-ordered uploads (host-fed replay, two-thread binding) A/B + the tests they touch
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
( timeout 900 python -m pytest tests -m gpu -x -q -k "threads or handoff or host or ouster or ring_split or replay" --durations=8 ) > $OUT/r05_b_pytest.log 2>&1
tail -15 $OUT/r05_b_pytest.log
for M in 0 1 2; do
  echo "=== LIODOM_UPLOAD_FLAGS=$M"; LIODOM_UPLOAD_FLAGS=$M timeout 300 python tools/dropin_rates.py hdl64 200 5 2>&1 | tail -4
done > $OUT/r05_b_dropin.txt 2>&1
for M in 1 2; do
  echo "=== LIODOM_UPLOAD_FLAGS=$M LIODOM_COPY_STREAM=1"; LIODOM_COPY_STREAM=1 LIODOM_UPLOAD_FLAGS=$M timeout 300 python tools/dropin_rates.py hdl64 200 5 2>&1 | tail -4
done >> $OUT/r05_b_dropin.txt 2>&1
echo "=== K=20 (driver-shaped), flags 1 / 0" >> $OUT/r05_b_dropin.txt
LIODOM_UPLOAD_FLAGS=1 timeout 300 python tools/dropin_rates.py hdl64 20 7 2>&1 | tail -4 >> $OUT/r05_b_dropin.txt
LIODOM_UPLOAD_FLAGS=0 timeout 300 python tools/dropin_rates.py hdl64 20 7 2>&1 | tail -4 >> $OUT/r05_b_dropin.txt
cat $OUT/r05_b_dropin.txt

#!/bin/bash
# (GPU) straight chain-mode replays with the predictor forced wrong (LIODOM_SPECULATE=2 / 6 / 7: both / the first / the finalising
# solve hand over as early as possible) against the replay without speculation, bit for bit, on three shapes.  usage: tools/spec_stress.sh [runs=4]
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
RUNS=${1:-4}
for SH in 16x900 vlp16 hdl64; do
  case $SH in hdl64) K=160;; vlp16) K=200;; *) K=240;; esac
  LIODOM_SPECULATE=0 python tools/spec_switches.py $SH $K /tmp/ss_ref.npy straight > /dev/null
  for M in 2 6 7; do
    BAD=0
    for i in $(seq $RUNS); do
      LIODOM_SPECULATE=$M python tools/spec_switches.py $SH $K /tmp/ss_x.npy straight > /dev/null
      python - <<'PY' || BAD=$((BAD+1))
import numpy as np, sys
a=np.load('/tmp/ss_ref.npy'); b=np.load('/tmp/ss_x.npy')
sys.exit(0 if np.array_equal(a.view(np.uint64), b.view(np.uint64)) else 1)
PY
    done
    echo "$SH: $K scans, LIODOM_SPECULATE=$M: $BAD of $RUNS replays differ from the replay without speculation"
  done
done

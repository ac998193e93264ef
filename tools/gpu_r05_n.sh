#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q --durations=6 ) > $OUT/r05_n_pytest.log 2>&1
tail -12 $OUT/r05_n_pytest.log
for W in hdl64 ouster128 vlp16; do timeout 300 python tools/stress_modes.py 3 $W 2>&1 | tail -1; done
for W in hdl64 ouster128 vlp16; do python3 tools/replay_trace.py $W 300; done

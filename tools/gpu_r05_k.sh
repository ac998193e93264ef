#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd $R
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests -m gpu -x -q -k "replay_resident_loop" 2>&1 | tail -1; done
( time timeout 1500 python -m pytest tests -m gpu -q --durations=6 ) > $OUT/r05_k_pytest.log 2>&1
tail -14 $OUT/r05_k_pytest.log
timeout 300 python tools/stress_modes.py 4 > $OUT/r05_k_stress.txt 2>&1; tail -2 $OUT/r05_k_stress.txt
timeout 300 python tools/stress_modes.py 3 vlp16 2>&1 | tail -1

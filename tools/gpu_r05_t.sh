#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
for i in 1 2 3; do timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -60 > $OUT/r05_t_suite_$i.txt; tail -1 $OUT/r05_t_suite_$i.txt; done
python tools/chain_hammer.py 40 vlp16 2>&1 | tail -3
for W in hdl64 ouster128 vlp16; do timeout 300 python tools/stress_modes.py 3 $W 2>&1 | tail -1; done
for W in hdl64 ouster128; do timeout 300 python tools/overlap_equal.py $W 200 2>&1 | tail -1; done
timeout 300 python tools/dropin_rates.py hdl64 200 3 2>&1 | tail -3

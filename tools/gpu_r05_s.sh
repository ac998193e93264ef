#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd $R
for i in 1 2 3; do ( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -2 ); done > $OUT/r05_s_pytest3.txt 2>&1
cat $OUT/r05_s_pytest3.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_s_bench_k20.json 2> $OUT/r05_s_bench_k20.stderr
python3 - <<PY
import json
d=json.load(open("$OUT/r05_s_bench_k20.json"))
print("K20 value", d["value"], d["value_spread"], "host_fed", d.get("host_fed_scans_per_s"), d["host_fed"].get("single_call_scans_per_s"), "two_thread", d.get("two_thread_scans_per_s"), "strict", d["strict_sync_scans_per_s"])
PY
for W in hdl64 vlp16 ouster128; do for i in 1 2 3; do timeout 300 python tools/stress_modes.py 2 $W 2>&1 | tail -1; done; done

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
tools/profile_round.sh r05_c > $OUT/r05_c_profile_round.log 2>&1
tail -5 $OUT/r05_c_pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_c_bench_k20.json 2> $OUT/r05_c_bench_k20.stderr
python3 - <<PY
import json
d=json.load(open("$OUT/r05_c_bench_k20.json"))
print("K20 value", d["value"], d["value_spread"], "host_fed", d.get("host_fed_scans_per_s"), d["host_fed"].get("single_call_scans_per_s"), "two_thread", d.get("two_thread_scans_per_s"), "strict", d["strict_sync_scans_per_s"])
PY
( time timeout 900 python -m pytest tests -m gpu -q --durations=8 ) > $OUT/r05_c_pytest_gpu_durations.log 2>&1
tail -14 $OUT/r05_c_pytest_gpu_durations.log

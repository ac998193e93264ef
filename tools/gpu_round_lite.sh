#!/bin/bash
# The bench legs of tools/profile_round.sh without the PMC passes: <tag>_bench{,_vlp16,_ouster128,_k20}.json + kernel trace (hdl64) + pytest log
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=6 --timeout 600 > $OUT/${TAG}_pytest_gpu.log 2>&1; tail -3 $OUT/${TAG}_pytest_gpu.log
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.stderr
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_k20.json 2> $OUT/${TAG}_bench_k20.stderr
for WLD in vlp16 ouster128; do timeout 600 python bench.py --workload $WLD --batched-streams 0 > $OUT/${TAG}_bench_$WLD.json 2> $OUT/${TAG}_bench_$WLD.stderr; done
( cd /tmp; rm -rf $OUT/prof_kt
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -- python3 $R/bench.py --workload hdl64 --steps 100 --warmup 20 --batched-streams 0 --no-cpu-baseline > /dev/null 2> $OUT/kt.stderr
  DB=$(find $OUT/prof_kt -name "*.db" | head -1)
  { echo "# $TAG: rocprofv3 --kernel-trace --stats -- python3 bench.py --workload hdl64 --steps 100 --warmup 20 --batched-streams 0 --no-cpu-baseline";
    echo "# legs: timed (pipelined) + HIP-event + async + strict + serial (+ host-fed, two-thread)";
    python3 $R/tools/rocprof_summary.py $DB; python3 $R/tools/rocprof_passes.py $DB; } > $OUT/${TAG}_bench_kernel_trace.txt
  rm -rf $OUT/prof_kt )
python3 - <<PY
import json
for suf in ("", "_k20", "_vlp16", "_ouster128"):
    d = json.load(open("$OUT/${TAG}_bench%s.json" % suf))
    print(suf or "hdl64", d["value"], d.get("value_spread"), "strict", d["strict_sync_scans_per_s"], "async", d.get("async_replay_scans_per_s"), "serial", d["serial_scans_per_s"], "host_fed", d["host_fed_scans_per_s"], d["host_fed"].get("single_call_scans_per_s"), "two_thread", d["two_thread_scans_per_s"], "batched", (d.get("batched") or {}).get("value"), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("reference_policy", {}).get("value"))
PY

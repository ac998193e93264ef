#!/bin/bash
# One GPU-box visit: selected tests + short bench + kernel trace (per-pass split).  usage: tools/gpu_step.sh <tag> [pytest -k expr]
set -u
TAG=${1:-step}
KEXPR=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
if [ -n "$KEXPR" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $OUT/${TAG}_pytest.log 2>&1; else timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest.log 2>&1; fi
tail -5 $OUT/${TAG}_pytest.log
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.stderr
python3 - <<PY
import json
try:
    d=json.load(open("$OUT/${TAG}_bench.json"))
    print("value", d["value"], "strict", d.get("strict_sync_scans_per_s"), "async", d.get("async_replay_scans_per_s"), "serial", d.get("serial_scans_per_s"))
    print("per_kernel_us", d["roofline"]["per_kernel_us"])
    b=d.get("batched")
    if b: print("batched", b["value"], b["ms_per_step"], b["roofline"]["per_kernel_us"])
    print("parity", d.get("parity"))
except Exception as e:
    print("bench parse failed", e); print(open("$OUT/${TAG}_bench.stderr").read()[-2000:])
PY
cd /tmp
rm -rf $OUT/prof_kt
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -- python3 $R/bench.py --steps 100 --warmup 20 --batched-streams 0 --no-cpu-baseline > /dev/null 2> $OUT/kt.stderr
DB=$(find $OUT/prof_kt -name "*.db" | head -1)
{ echo "# $TAG: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 20 --batched-streams 0 --no-cpu-baseline";
  python3 $R/tools/rocprof_summary.py $DB; python3 $R/tools/rocprof_passes.py $DB; } > $OUT/${TAG}_kernel_trace.txt
rm -rf $OUT/prof_kt
head -16 $OUT/${TAG}_kernel_trace.txt; tail -4 $OUT/${TAG}_kernel_trace.txt

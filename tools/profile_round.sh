#!/bin/bash
# Produces the per-round measurement artefacts on the GPU box (run through gpurun):
#   gpurun_out/<tag>_bench.json, <tag>_bench_kernel_trace.txt, <tag>_pmc_traffic.{txt,json}, <tag>_pytest_gpu.log
# usage: tools/profile_round.sh r01_e
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
if [ "${SKIP_PYTEST:-0}" != "1" ]; then timeout 1200 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest_gpu.log 2>&1; fi
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.stderr
if [ "${OTHER_WORKLOADS:-1}" = "1" ]; then
  for WLD in vlp16 ouster128; do
    timeout 600 python bench.py --workload $WLD --batched-streams 0 > $OUT/${TAG}_bench_${WLD}.json 2> $OUT/${TAG}_bench_${WLD}.stderr
    ( cd /tmp; rm -rf $OUT/prof_kt_$WLD
      timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt_$WLD -- python3 $R/bench.py --workload $WLD --steps 100 --warmup 20 --batched-streams 0 --no-cpu-baseline > /dev/null 2> $OUT/kt_$WLD.stderr
      DBW=$(find $OUT/prof_kt_$WLD -name "*.db" | head -1)
      { echo "# $TAG: rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $WLD --steps 100 --warmup 20 --batched-streams 0 --no-cpu-baseline";
        python3 $R/tools/rocprof_summary.py $DBW; } > $OUT/${TAG}_bench_${WLD}_kernel_trace.txt
      rm -rf $OUT/prof_kt_$WLD )
  done
fi
cd /tmp
rm -rf $OUT/prof_kt $OUT/prof_fetch $OUT/prof_write
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -- python3 $R/bench.py --steps 100 --warmup 20 --batched-streams 0 --no-cpu-baseline > $OUT/${TAG}_kt_bench.json 2> $OUT/kt.stderr
DB=$(find $OUT/prof_kt -name "*.db" | head -1)
{ echo "# $TAG: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 20 --batched-streams 0 --no-cpu-baseline";
  echo "# 64x1800, R=8, epr=10, P=20, 1 stream; legs: timed (pipelined) + HIP-event + async + serial => 4 x 120 scans";
  python3 $R/tools/rocprof_summary.py $DB; } > $OUT/${TAG}_bench_kernel_trace.txt
# (--pmc serialises kernels across streams: the flag-based stream dependencies cannot make progress there; use the event path)
export LIODOM_PIPE_FLAGS=0
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch -- python3 $R/bench.py --steps 40 --warmup 20 --batched-streams 16 --no-cpu-baseline > /dev/null 2> $OUT/fetch.stderr
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write -- python3 $R/bench.py --steps 40 --warmup 20 --batched-streams 16 --no-cpu-baseline > /dev/null 2> $OUT/write.stderr
FD=$(dirname $(find $OUT/prof_fetch -name "*.db" | head -1))
WD=$(dirname $(find $OUT/prof_write -name "*.db" | head -1))
{ echo "# $TAG: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 40 --warmup 20 --batched-streams 16 --no-cpu-baseline";
  echo "# small grid = 1 stream (headline), large grid = 16 lock-step streams. Per-launch averages.";
  python3 $R/tools/pmc_summary.py $FD $WD $OUT/${TAG}_pmc_traffic.json; } > $OUT/${TAG}_pmc_traffic.txt
unset LIODOM_PIPE_FLAGS
rm -rf $OUT/prof_kt $OUT/prof_fetch $OUT/prof_write
tail -3 $OUT/${TAG}_pytest_gpu.log 2>/dev/null; cat $OUT/${TAG}_bench.json | head -c 600; echo; head -14 $OUT/${TAG}_bench_kernel_trace.txt; head -30 $OUT/${TAG}_pmc_traffic.txt

#!/bin/bash
# Produces the per-round measurement artefacts on the GPU box (run through gpurun):
#   gpurun_out/<tag>_bench.json, <tag>_bench_kernel_trace.txt, <tag>_pmc_traffic_<workload>.{txt,json}, <tag>_sq.txt, <tag>_pytest_gpu.log
# usage: tools/profile_round.sh r03_e        (env: SKIP_PYTEST=1, OTHER_WORKLOADS=0, SKIP_SQ=1)
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
if [ "${SKIP_PYTEST:-0}" != "1" ]; then timeout 1800 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest_gpu.log 2>&1; tail -3 $OUT/${TAG}_pytest_gpu.log; fi
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.stderr
WLS="hdl64"
if [ "${OTHER_WORKLOADS:-1}" = "1" ]; then WLS="hdl64 vlp16 ouster128"; fi
for WLD in $WLS; do
  SUF="_$WLD"; if [ $WLD = hdl64 ]; then SUF=""; fi
  if [ $WLD != hdl64 ]; then timeout 600 python bench.py --workload $WLD --batched-streams 0 > $OUT/${TAG}_bench${SUF}.json 2> $OUT/${TAG}_bench${SUF}.stderr; fi
  ( cd /tmp; rm -rf $OUT/prof_kt
    timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -- python3 $R/bench.py --workload $WLD --steps 100 --warmup 20 --batched-streams 0 --no-cpu-baseline > /dev/null 2> $OUT/kt_$WLD.stderr
    DB=$(find $OUT/prof_kt -name "*.db" | head -1)
    { echo "# $TAG: rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $WLD --steps 100 --warmup 20 --batched-streams 0 --no-cpu-baseline";
      echo "# legs: timed (pipelined) + HIP-event + async + strict + serial (+ host-fed, two-thread)";
      python3 $R/tools/rocprof_summary.py $DB; python3 $R/tools/rocprof_passes.py $DB; } > $OUT/${TAG}_bench${SUF}_kernel_trace.txt
    rm -rf $OUT/prof_kt
    # (--pmc serialises kernels across streams: the flag-based stream dependencies cannot make progress there; use the event path)
    export LIODOM_PIPE_FLAGS=0
    for C in FETCH_SIZE WRITE_SIZE; do
      rm -rf $OUT/pmc1_$C $OUT/pmc16_$C
      timeout 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc1_$C -- python3 $R/tools/workload_run.py $WLD 1 20 > /dev/null 2> $OUT/pmc1_$C.err
      if [ $WLD = hdl64 ]; then
        timeout 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc16_$C -- python3 $R/tools/workload_run.py $WLD 16 20 > /dev/null 2> $OUT/pmc16_$C.err
        timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc256_$C -- python3 $R/tools/workload_run.py $WLD 256 20 > /dev/null 2> $OUT/pmc256_$C.err
      fi
    done
    unset LIODOM_PIPE_FLAGS
    case $WLD in vlp16) PF=10;; ouster128) PF=30;; *) PF=20;; esac
    { echo "# $TAG: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/workload_run.py $WLD <streams> 20   (window pre-filled first: the $PF pre-fill scans' launches are dropped; LIODOM_PIPE_FLAGS=0)";
      if [ $WLD = hdl64 ]; then PMC_STEADY_SCANS=20 python3 $R/tools/pmc_summary.py $WLD $OUT/${TAG}_pmc_traffic_$WLD.json $PF $OUT/pmc1_FETCH_SIZE $OUT/pmc1_WRITE_SIZE 16 $OUT/pmc16_FETCH_SIZE $OUT/pmc16_WRITE_SIZE 256 $OUT/pmc256_FETCH_SIZE $OUT/pmc256_WRITE_SIZE;
      else PMC_STEADY_SCANS=20 python3 $R/tools/pmc_summary.py $WLD $OUT/${TAG}_pmc_traffic_$WLD.json $PF $OUT/pmc1_FETCH_SIZE $OUT/pmc1_WRITE_SIZE; fi; } > $OUT/${TAG}_pmc_traffic_$WLD.txt
    rm -rf $OUT/pmc1_FETCH_SIZE $OUT/pmc1_WRITE_SIZE $OUT/pmc16_FETCH_SIZE $OUT/pmc16_WRITE_SIZE $OUT/pmc256_FETCH_SIZE $OUT/pmc256_WRITE_SIZE )
done
if [ "${SKIP_SQ:-0}" != "1" ]; then SQ_STREAMS=64 $R/tools/pmc_sq.sh $TAG > /dev/null 2>&1; fi
cat $OUT/${TAG}_bench.json | head -c 900; echo; head -14 $OUT/${TAG}_bench_kernel_trace.txt; head -24 $OUT/${TAG}_pmc_traffic_hdl64.txt

#!/bin/bash
# serialised per-kernel durations (+ one PMC group) of the lock-step batch for the product library and variants.
# usage: PROBE_GREP="k_hash" tools/r06_probe.sh [streams] [variant|old|noincr ...]
S=${1:-256}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
K=${PROBE_SCANS:-8}; SKIP=$(python3 -c "print(20.0/(20+$K))")
export LIODOM_PIPE_FLAGS=0
G="${PROBE_GROUP:-SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES}"
for V in product "$@"; do
  rm -rf /tmp/pq
  if [ "$V" = product ]; then timeout 600 rocprofv3 --kernel-trace --pmc $G -d /tmp/pq -- python3 $R/tools/workload_run.py hdl64 $S $K > /dev/null 2>/tmp/pq.err
  elif [ "$V" = old ]; then LIODOM_KNN8=0 timeout 600 rocprofv3 --kernel-trace --pmc $G -d /tmp/pq -- python3 $R/tools/workload_run.py hdl64 $S $K > /dev/null 2>/tmp/pq.err
  elif [ "$V" = noincr ]; then LIODOM_HASH_INCR=0 timeout 600 rocprofv3 --kernel-trace --pmc $G -d /tmp/pq -- python3 $R/tools/workload_run.py hdl64 $S $K > /dev/null 2>/tmp/pq.err
  else timeout 600 rocprofv3 --kernel-trace --pmc $G -d /tmp/pq -- python3 $R/tools/variant_run.py $V $R/tools/workload_run.py hdl64 $S $K > /dev/null 2>/tmp/pq.err; fi
  echo "== $V [$G]"; python3 $R/tools/pmc_passes.py $SKIP /tmp/pq | grep -E "${PROBE_GREP:-k_}"
done

#!/bin/bash
# kernel-trace timing (per-pass split) at VAR_STREAMS (64) lock-step streams for each variant library ("product" = the product library).
# usage: tools/variant_time.sh name...
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
K=${VAR_SCANS:-6}
for V in "$@"; do
  rm -rf /tmp/pv
  if [ "$V" = product ]; then
    rocprofv3 --kernel-trace -d /tmp/pv -- python3 $R/tools/workload_run.py hdl64 ${VAR_STREAMS:-64} $K > /dev/null 2>/tmp/pv.err
  else
    rocprofv3 --kernel-trace -d /tmp/pv -- python3 $R/tools/variant_run.py $V $R/tools/workload_run.py hdl64 ${VAR_STREAMS:-64} $K > /dev/null 2>/tmp/pv.err
  fi
  echo "== $V (${VAR_STREAMS:-64} streams)"
  python3 $R/tools/pmc_passes.py $(python3 -c "print(20.0/(20+$K))") /tmp/pv | grep -E "${VAR_GREP:-k_knn|k_line_gate|k_hash_build|k_ring|k_classify|k_lm_solve}"
done

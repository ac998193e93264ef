#!/bin/bash
# SQ-counter passes (rocprofv3 --pmc, one group per run) for the lock-step batch (64 streams) and one stream.
# usage: tools/pmc_sq.sh <tag>   -> gpurun_out/<tag>_sq.txt
set -u
TAG=${1:-sq}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
export LIODOM_PIPE_FLAGS=0      # --pmc serialises kernels across streams
cd /tmp
rocprofv3 -L > $OUT/${TAG}_counters_available.txt 2>&1 || true
G1="SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
G3="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
G4="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM"
i=0
DIRS64=""; DIRS1=""
for G in "$G1" "$G2" "$G3" "$G4"; do
  i=$((i+1))
  rm -rf $OUT/sq64_$i $OUT/sq1_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $G -d $OUT/sq64_$i -- python3 $R/tools/workload_run.py hdl64 ${SQ_STREAMS:-64} 26 > /dev/null 2> $OUT/sq64_$i.err
  timeout 300 rocprofv3 --kernel-trace --pmc $G -d $OUT/sq1_$i -- python3 $R/tools/workload_run.py hdl64 1 46 > /dev/null 2> $OUT/sq1_$i.err
  DIRS64="$DIRS64 $OUT/sq64_$i"; DIRS1="$DIRS1 $OUT/sq1_$i"
done
{ echo "# $TAG: rocprofv3 --kernel-trace --pmc <group> -- python3 tools/workload_run.py hdl64 ${SQ_STREAMS:-64} 26   (lock-step streams, per-launch averages; LIODOM_PIPE_FLAGS=0)";
  echo "# groups: [$G1] [$G2] [$G3] [$G4]   SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles summed over waves (MI355X_MICROARCH.md)";
  python3 $R/tools/pmc_generic.py $DIRS64;
  echo; echo "# one stream: python3 tools/workload_run.py hdl64 1 46";
  python3 $R/tools/pmc_generic.py $DIRS1; } > $OUT/${TAG}_sq.txt
for d in $DIRS64 $DIRS1; do rm -rf $d; done
grep -E "k_knn|k_line_gate|k_lm_solve|k_ring_extract" $OUT/${TAG}_sq.txt | cut -c1-400

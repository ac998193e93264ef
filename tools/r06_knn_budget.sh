#!/bin/bash
# Round 6, first GPU visit: where the batched kNN spends its time.  usage: tools/r06_knn_budget.sh [streams=256]
set -u
S=${1:-256}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
K=6; SKIP=0.77      # 20 pre-fill scans + 6 steady ones: the first 77 % of every kernel's dispatches are dropped
{
echo "# r06 knn budget: tools/workload_run.py hdl64 $S $K (P = 20 pre-fill scans dropped), per-pass split by dispatch order"
cd /tmp
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace -d /tmp/kt -- python3 $R/tools/workload_run.py hdl64 $S $K > /dev/null 2> $OUT/kt_base.err
echo "## kernel trace (durations, us)"; python3 $R/tools/pmc_passes.py $SKIP /tmp/kt
export LIODOM_PIPE_FLAGS=0
i=0
for G in "SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  timeout 600 rocprofv3 --kernel-trace --pmc $G -d /tmp/pm$i -- python3 $R/tools/workload_run.py hdl64 $S $K > /dev/null 2> $OUT/pm$i.err
  echo "## pmc group $i: $G"; python3 $R/tools/pmc_passes.py $SKIP /tmp/pm$i | grep -E "k_knn|k_line_gate|k_hash_build|k_ring|k_classify|k_lm_solve"
done
unset LIODOM_PIPE_FLAGS
} > $OUT/r06_knn_budget_$S.txt 2>&1
cat $OUT/r06_knn_budget_$S.txt | cut -c1-600

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
{ VAR_STREAMS=64 tools/variant_time.sh base pk base pk; VAR_STREAMS=256 tools/variant_time.sh base pk; } > $OUT/r05_p_pk.txt 2>&1
cat $OUT/r05_p_pk.txt

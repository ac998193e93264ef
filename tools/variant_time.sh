#!/bin/bash
# kernel-trace timing of k_knn / k_line_gate at 64 lock-step streams for each variant library. usage: tools/variant_time.sh name...
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for V in "$@"; do
  rm -rf /tmp/pv
  rocprofv3 --kernel-trace --stats -d /tmp/pv -- python3 $R/tools/variant_run.py $V $R/tools/workload_run.py hdl64 ${VAR_STREAMS:-64} 26 > /dev/null 2>/tmp/pv.err
  DB=$(find /tmp/pv -name "*.db" | head -1)
  python3 - <<PY
import sqlite3
db=sqlite3.connect("$DB")
out=[]
for kern in ("k_knn","k_line_gate","k_lm_solve","k_ring_extract","k_ring_scatter","k_classify","k_hash_build"):
    rows=db.execute("select duration from kernels where name like ? order by start",("%"+kern+"%",)).fetchall()
    rows=[r[0]/1e3 for r in rows][len(rows)//2:]
    if rows: out.append("%s %.1f"%(kern,sum(rows)/len(rows)))
print("$V:", "  ".join(out))
PY
done

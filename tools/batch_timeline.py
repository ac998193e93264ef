"""Timeline of one steady-state step of the pipelined lock-step batch from a rocprofv3 --kernel-trace .db: every kernel with start / end
relative to the step's first kernel, its queue, and how much of its duration another queue's kernel was running beside it.
usage: python tools/batch_timeline.py <results.db> [step_from_the_end=3]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cols = [c[1] for c in db.execute("pragma table_info(kernels)").fetchall()]
qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
rows = db.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")).fetchall()
short = lambda n: n.split("(")[0].replace("void ", "").replace("liodom_dev::", "")
# steps are delimited by k_ring_extract launches
ext = [i for i, r in enumerate(rows) if "k_ring_extract" in r[0]]
i0, i1 = ext[-back - 1], ext[-back]
t0 = rows[i0][1]
step = [r for r in rows if r[1] >= t0 and r[1] < rows[i1][1]]
print("step of %.1f us (k_ring_extract to k_ring_extract), %d kernels; queue column: %s" % ((rows[i1][1] - t0) / 1e3, len(step), qcol))
for name, st, en, q in step:
    ov = 0
    for n2, s2, e2, q2 in rows:
        if q2 != q and s2 < en and e2 > st:
            ov += min(en, e2) - max(st, s2)
    print("%-28s q %-6s start %8.1f  end %8.1f  dur %7.1f  beside another queue's kernel %6.1f us" % (short(name)[:28], q, (st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, ov / 1e3))

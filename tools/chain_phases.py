"""Per-scan phases of the odometry chain from a rocprofv3 kernel trace (sqlite .db) of tools/replay_trace.py: for the last N scans,
the intervals between the ends of the chain's links.  usage: python tools/chain_phases.py <db> [N]"""
import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = db.execute("select name, start, end from kernels order by start").fetchall()
def short(n):
    return n.split("(")[0].replace("void ", "").replace("liodom_dev::", "")
ev = [(short(n), a, b) for n, a, b in rows]
k0 = [e for e in ev if e[0].startswith("k_knn<256, false")]
k1 = [e for e in ev if e[0].startswith("k_knn<256, true")]
lm = [e for e in ev if e[0].startswith("k_lm_solve")]
fin = [e for e in ev if e[0] == "k_rebuild_fin"]
scans = []
for a in k0[-N - 1:-1]:
    # the launches of this scan: first k1 starting after a's start, the two solves ending after a's start
    nk0 = [e for e in k0 if e[1] > a[1]]
    nxt = nk0[0] if nk0 else None
    b1 = [e for e in k1 if e[1] >= a[1] and (nxt is None or e[1] < nxt[1])]
    sol = [e for e in lm if e[2] > a[1] and (nxt is None or e[2] <= nxt[2] + 1)][:2]
    fn = [e for e in fin if e[1] >= a[1] and (nxt is None or e[1] < nxt[2])]
    if len(b1) != 1 or len(sol) != 2 or nxt is None:
        continue
    s0, s1 = sol
    end = max(s1[2], fn[0][2]) if fn else s1[2]
    scans.append(dict(k0=(a[2] - a[1]), s0_after_k0=(s0[2] - a[2]), s0_start_vs_k0_start=(s0[1] - a[1]), k1_after_s0=(b1[0][2] - s0[2]),
                      s1_after_k1=(s1[2] - b1[0][2]), tail_after_s1=(end - s1[2]), gap_to_next_k0=(nxt[1] - end), period=(nxt[1] - a[1])))
print("scans analysed:", len(scans))
for key in ("k0", "s0_start_vs_k0_start", "s0_after_k0", "k1_after_s0", "s1_after_k1", "tail_after_s1", "gap_to_next_k0", "period"):
    x = np.array([s[key] for s in scans]) / 1e3
    print("%-22s mean %7.2f  median %7.2f  p90 %7.2f us" % (key, x.mean(), np.median(x), np.percentile(x, 90)))

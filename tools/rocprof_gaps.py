"""Gaps on the odometry chain of the first (per-scan synchronous) leg of a bench.py kernel trace: time from the end of
each k_knn / k_lm_solve dispatch to the start of the next, by position in the scan (kNN0->LM0, LM0->kNN1, kNN1->LM1, LM1->next kNN0)."""
import sqlite3
import statistics as st
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, name from kernels where name like '%k_knn%' or name like '%k_lm_solve%' order by start").fetchall()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
first = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = rows[4 * first:4 * (first + n)]          # skip the pre-fill, take n scans of the leg
names = ["kNN0->LM0", "LM0->kNN1", "kNN1->LM1", "LM1->next kNN0"]
g = [[] for _ in range(4)]
d = [[] for _ in range(4)]
for i in range(len(rows) - 1):
    g[i % 4].append((rows[i + 1][0] - rows[i][1]) / 1e3)
    d[i % 4].append((rows[i][1] - rows[i][0]) / 1e3)
for k in range(4):
    print("%-16s gap median %.2f us (mean %.2f)   kernel before it: median %.2f us" % (names[k], st.median(g[k]), sum(g[k]) / len(g[k]), st.median(d[k])))
print("scan period (kNN0 start to next kNN0 start): median %.2f us" % st.median([(rows[i + 4][0] - rows[i][0]) / 1e3 for i in range(0, len(rows) - 4, 4)]))

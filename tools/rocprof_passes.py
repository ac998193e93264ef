"""Per-pass durations of k_knn / k_lm_solve from a rocprofv3 kernel-trace .db: dispatches of each kernel in start
order alternate between the scan's first and second pass."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
for kern in ("k_knn", "k_lm_solve"):
    rows = db.execute("select start, duration from kernels where name like ? order by start", ("%" + kern + "%",)).fetchall()
    # bench.py runs five legs of equal length over the same scans (timed pipelined, HIP-event, asynchronous, strictly
    # synchronous, serial); the first one is the headline mode: its steady-state part (after pre-fill + warm-up)
    if len(rows) % 10 == 0 and len(rows) >= 100:
        leg = len(rows) // 5
        rows = rows[leg // 2 + (leg // 2) % 2:leg]
    else:
        rows = rows[-400:]
    if len(rows) % 2:
        rows = rows[1:]
    a = [r[1] / 1e3 for r in rows[0::2]]
    b = [r[1] / 1e3 for r in rows[1::2]]
    import statistics as st
    print("%-12s even dispatches: avg %.2f us (median %.2f, max %.2f)   odd dispatches: avg %.2f us (median %.2f, max %.2f)   n=%d" %
          (kern, sum(a) / len(a), st.median(a), max(a), sum(b) / len(b), st.median(b), max(b), len(a)))
# gaps on the odometry chain: end of one kernel to the start of the next among k_knn / k_lm_solve
rows = db.execute("select start, end, name from kernels where name like '%k_knn%' or name like '%k_lm_solve%' order by start").fetchall()[-400:]
gaps = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(len(rows) - 1)]
gaps = [g for g in gaps if g < 100]
print("gaps between consecutive odometry kernels: avg %.2f us, median %.2f" % (sum(gaps) / len(gaps), sorted(gaps)[len(gaps) // 2]))

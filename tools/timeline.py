"""Prints the kernel timeline (start / end in us relative to the first kernel shown) of the last scans in a rocprofv3 kernel-trace db.
usage: python tools/timeline.py <db> [n_kernels]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall() if False else None
try:
    rows = db.execute("select name, start, end, queue_id from kernels order by start").fetchall()
except Exception:
    rows = db.execute("select name, start, end, 0 from kernels order by start").fetchall()
rows = rows[-n:]
t0 = rows[0][1]
for name, a, b, q in rows:
    nm = name.split("(")[0].replace("void ", "").replace("liodom_dev::", "")
    print("%-34s q%-3s %9.2f -> %9.2f  (%6.2f)" % (nm[:34], q, (a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3))

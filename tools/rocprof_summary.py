"""Summarise a rocprofv3 results .db (kernel trace) as a per-kernel table (name, calls, avg/min/max us, %)."""
import sqlite3
import sys


def summarise(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    out = ["%-60s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct")]
    for name, n, s, a, mn, mx in rows:
        short = name.split("(")[0].replace("void ", "")[-60:]
        out.append("%-60s %7d %12.1f %10.2f %10.2f %10.2f %6.1f" % (short, n, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    span = db.execute("select min(start), max(end) from kernels").fetchone()
    out.append("kernel time total %.1f us; first-start to last-end span %.1f us" % (tot / 1e3, (span[1] - span[0]) / 1e3))
    return "\n".join(out)


if __name__ == "__main__":
    print(summarise(sys.argv[1]))

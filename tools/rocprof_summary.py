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
    # k_lm_solve by launch shape: in chain mode (round 5) a solve launch holds the solving workgroups alone and is RESIDENT while the
    # kNN pass before it still runs — its duration includes that wait; the four-launch chain's launches carry the streamed rebuild's
    # workgroups (larger grids) and start when the pass has ended.  Only the latter's duration is the kernel's work.
    try:
        cols = [c[1] for c in db.execute("pragma table_info(kernels)").fetchall()]
        gcol = "grid_size" if "grid_size" in cols else ("grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None))
        if gcol:
            shapes = db.execute("select %s, count(*), avg(duration), min(duration), max(duration) from kernels where name like '%%k_lm_solve%%' "
                                "group by %s order by %s" % (gcol, gcol, gcol)).fetchall()
            if len(shapes) > 1:
                out.append("k_lm_solve by grid (work-items): " + "; ".join("%d: %d launches, avg %.2f us (min %.2f, max %.2f)" % (g, n, a / 1e3, mn / 1e3, mx / 1e3)
                                                                         for g, n, a, mn, mx in shapes))
    except Exception as ex:       # (schema differences between rocprofv3 versions: the table above is what matters)
        out.append("k_lm_solve by grid: not available (%s)" % ex)
    span = db.execute("select min(start), max(end) from kernels").fetchone()
    out.append("kernel time total %.1f us; first-start to last-end span %.1f us" % (tot / 1e3, (span[1] - span[0]) / 1e3))
    return "\n".join(out)


if __name__ == "__main__":
    print(summarise(sys.argv[1]))

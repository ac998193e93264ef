"""Per-kernel averages of the PMC counters in a rocprofv3 --pmc results .db (largest grid of each kernel only)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
pmc = [t for t in tabs if t.startswith("counters_collection") or t == "counters_collection"]
view = "counters_collection" if "counters_collection" in tabs else pmc[0]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % view)]
rows = db.execute("select kernel_name, grid_size, counter_name, avg(value), count(*) from %s group by kernel_name, grid_size, counter_name" % view).fetchall()
best = {}
for name, grid, cname, val, n in rows:
    short = name.split("(")[0].replace("void ", "").replace("liodom_dev::", "")
    key = short
    if key not in best or grid > best[key][0]:
        best[key] = (grid, {})
    if grid == best[key][0]:
        best[key][1][cname] = (val, n)
for k, (grid, cs) in sorted(best.items()):
    print("%-28s grid %9d  " % (k[:28], grid) + "  ".join("%s=%.4g" % (c, v[0]) for c, v in sorted(cs.items())))

"""Per-kernel averages of arbitrary rocprofv3 --pmc counters from one or more result dirs."""
import glob, sqlite3, sys
rows = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*.db", recursive=True):
        db = sqlite3.connect(f)
        for k, g, c, n, v in db.execute("select kernel_name, grid_size, counter_name, count(*), avg(value) from counters_collection group by kernel_name, grid_size, counter_name"):
            if "liodom_dev" not in k: continue
            rows.setdefault((k.split("(")[0].split("<")[0].replace("void ", "").replace("liodom_dev::", ""), g), {})[c] = (n, v)
for key in sorted(rows):
    print("%-18s grid %9d  " % key + "  ".join("%s=%.4g" % (c, v) for c, (n, v) in sorted(rows[key].items())))

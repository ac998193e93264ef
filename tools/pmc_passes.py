"""rocprofv3 --pmc result dirs -> per-kernel counter averages with the two passes of a scan kept apart: the dispatches of k_knn,
k_line_gate and k_lm_solve alternate (first / second pass) in dispatch order.  Also works on --kernel-trace dirs (durations).
usage: pmc_passes.py <skip_fraction> <dir> [<dir> ...]      skip_fraction: leading share of every kernel's dispatches to drop (pre-fill)"""
import glob, sqlite3, sys
skip = float(sys.argv[1])
SPLIT = ("k_knn", "k_knn8", "k_line_gate", "k_lm_solve")
acc = {}
dur = {}
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*.db", recursive=True):
        db = sqlite3.connect(f)
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" in tabs:
            per = {}
            for k, g, disp, c, v in db.execute("select kernel_name, grid_size, dispatch_id, counter_name, value from counters_collection"):
                if "liodom_dev" not in k: continue
                name = k.split("(")[0].split("<")[0].replace("void ", "").replace("liodom_dev::", "")
                per.setdefault((name, c), {}).setdefault(disp, 0.0)
                per[(name, c)][disp] += v
            for (name, c), dd in per.items():
                vals = [dd[i] for i in sorted(dd)]
                vals = vals[int(len(vals) * skip):]
                if name in SPLIT:
                    if len(vals) % 2: vals = vals[1:]
                    acc.setdefault(name + " pass0", {})[c] = sum(vals[0::2]) / max(1, len(vals[0::2]))
                    acc.setdefault(name + " pass1", {})[c] = sum(vals[1::2]) / max(1, len(vals[1::2]))
                else:
                    acc.setdefault(name, {})[c] = sum(vals) / max(1, len(vals))
        if "kernels" in tabs and not dur:
            per = {}
            for k, st, du in db.execute("select name, start, duration from kernels order by start"):
                if "liodom_dev" not in k: continue
                name = k.split("(")[0].split("<")[0].replace("void ", "").replace("liodom_dev::", "")
                per.setdefault(name, []).append(du / 1e3)
            for name, vals in per.items():
                vals = vals[int(len(vals) * skip):]
                if name in SPLIT:
                    if len(vals) % 2: vals = vals[1:]
                    dur[name + " pass0"] = (sum(vals[0::2]) / max(1, len(vals[0::2])), len(vals[0::2]))
                    dur[name + " pass1"] = (sum(vals[1::2]) / max(1, len(vals[1::2])), len(vals[1::2]))
                else:
                    dur[name] = (sum(vals) / max(1, len(vals)), len(vals))
for name in sorted(set(acc) | set(dur)):
    line = "%-22s" % name
    if name in dur: line += " dur_us=%.1f (n=%d)" % dur[name]
    line += "  " + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(acc.get(name, {}).items()))
    print(line)

"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes) into the per-workload traffic profile bench.py reads.

usage: pmc_summary.py <workload> <out.json> <skip_scans> <single_fetch_dir> <single_write_dir> [<batched_streams> <batched_fetch_dir> <batched_write_dir>]...

skip_scans: the workload's pre-fill (tools/workload_run.py runs P scans into an empty window first): of every (kernel, grid)
the first skip_scans / total_scans share of the launches — in dispatch order — is dropped, so that the averages are those of the
steady state bench.py times.  Several batched groups may follow (e.g. 16 and 256 lock-step streams); bench.py uses the one whose
stream count equals its own and never scales a measurement to another stream count.

FETCH_SIZE / WRITE_SIZE are reported in KiB; per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on
gfx950 counts 128-B requests at 64 B, i.e. reads exactly half of a wide coalesced stream — `fetch_corrected` doubles it
(calibration in these traces: k_classify reads 16 B/point = 1843 KB and reports ~925 KB).  WRITE_SIZE is uncalibrated
(k_classify writes 1 B/point = 113 KB and reports ~113 KB, so it is taken as is).  A kernel launched with several grid
sizes in one run (k_knn / k_lm_solve: first and second pass of a scan) is averaged over all its launches; the single-stream
run and the lock-step run are kept apart, so a row is always selected by launch shape."""
import glob
import json
import sqlite3
import sys


SKIP_FRACTION = 0.0     # share of every (kernel, grid)'s launches that belongs to the pre-fill (dispatch order)


def load(dirname, counter):
    f = glob.glob(dirname + "/**/*.db", recursive=True)[0]
    db = sqlite3.connect(f)
    try:
        rows = db.execute("select kernel_name, grid_size, dispatch_id, value from counters_collection where counter_name=? "
                          "order by dispatch_id", (counter,)).fetchall()
    except sqlite3.Error:
        rows = db.execute("select kernel_name, grid_size, rowid, value from counters_collection where counter_name=? order by rowid",
                          (counter,)).fetchall()
    per = {}
    for k, g, d, v in rows:
        if "liodom_dev" not in k:
            continue
        name = k.split("(")[0].split("<")[0].replace("void ", "").replace("liodom_dev::", "")
        per.setdefault((name, g), {}).setdefault(d, 0.0)
        per[(name, g)][d] += v                      # (one row per counter instance / XCC: summed per dispatch)
    out = {}
    for (name, g), disp in per.items():
        vals = [disp[d] for d in sorted(disp)]
        drop = int(round(len(vals) * SKIP_FRACTION))
        vals = vals[drop:] if len(vals) > drop else vals
        out.setdefault(name, []).append((g, len(vals), sum(vals) / max(len(vals), 1)))
    return out


def merge(fetch_dir, write_dir, title):
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    res = {}
    print("# " + title)
    print("%-18s %-26s %8s %14s %18s %14s %16s" % ("kernel", "grids (launches)", "launches", "FETCH_SIZE KiB", "fetch_corrected KiB", "WRITE_SIZE KiB", "HBM B / launch"))
    for name in sorted(fetch):
        nf = sum(n for _, n, _ in fetch[name])
        fv = sum(n * v for _, n, v in fetch[name]) / max(nf, 1)
        w = write.get(name, [])
        nw = sum(n for _, n, _ in w)
        wv = sum(n * v for _, n, v in w) / max(nw, 1)
        grids = " ".join("%d(%d)" % (g, n) for g, n, _ in sorted(fetch[name]))
        hbm = int((2 * fv + wv) * 1024)
        print("%-18s %-26s %8d %14.1f %18.1f %14.1f %16d" % (name, grids[:26], nf, fv, 2 * fv, wv, hbm))
        res[name] = {"launches": nf, "grids": sorted(g for g, _, _ in fetch[name]), "fetch_kib_raw": fv, "fetch_kib_corrected": 2 * fv,
                     "write_kib": wv, "hbm_bytes_per_launch": hbm}
    return res


if __name__ == "__main__":
    workload, out, skip = sys.argv[1], sys.argv[2], int(sys.argv[3])
    total = skip + int(__import__("os").environ.get("PMC_STEADY_SCANS", "20"))
    SKIP_FRACTION = skip / float(total)
    prof = {"workload": workload, "prefill_scans_dropped": skip, "steady_scans": total - skip,
            "single": merge(sys.argv[4], sys.argv[5], "%s, one stream, steady state (per-launch averages over the launches behind the %d pre-fill scans)" % (workload, skip))}
    rest = sys.argv[6:]
    groups = []
    while len(rest) >= 3:
        S = int(rest[0])
        groups.append({"streams": S, "kernels": merge(rest[1], rest[2], "%s, %d lock-step streams, steady state (per-launch averages, whole launch)" % (workload, S))})
        rest = rest[3:]
    if groups:
        prof["batched"] = groups[-1]               # (the largest stream count given last: what bench.py's batched leg runs)
        prof["batched_groups"] = groups
    json.dump(prof, open(out, "w"), indent=1)

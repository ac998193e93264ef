"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes) per kernel and grid size.
FETCH_SIZE / WRITE_SIZE are reported in KiB; per /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
FETCH_SIZE on gfx950 counts 128-B requests at 64 B, i.e. reads exactly half of a wide coalesced
stream — `fetch_corrected` doubles it (calibration in this very trace: k_classify reads 16 B/point =
1843 KB and reports 925 KB).  WRITE_SIZE is uncalibrated (k_classify writes 1 B/point = 113 KB and
reports 112.75 KB, so it is taken as is)."""
import glob
import json
import sqlite3
import sys


def load(dirname, counter):
    f = glob.glob(dirname + "/*.db")[0]
    db = sqlite3.connect(f)
    rows = db.execute("select kernel_name, grid_size, count(*), avg(value) from counters_collection "
                      "where counter_name=? group by kernel_name, grid_size", (counter,)).fetchall()
    return {(k.split("(")[0].split("<")[0].replace("void ", "").replace("liodom_dev::", ""), g): (n, v) for k, g, n, v in rows if "liodom_dev" in k}


if __name__ == "__main__":
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    print("%-18s %10s %8s %14s %18s %14s" % ("kernel", "grid", "launches", "FETCH_SIZE KiB", "fetch_corrected KiB", "WRITE_SIZE KiB"))
    for key in sorted(fetch):
        n, fv = fetch[key]
        wv = write.get(key, (0, 0.0))[1]
        print("%-18s %10d %8d %14.1f %18.1f %14.1f" % (key[0], key[1], n, fv, 2 * fv, wv))
        out.setdefault(key[0], []).append({"grid": key[1], "launches": n, "fetch_kib_raw": fv, "fetch_kib_corrected": 2 * fv,
                                           "write_kib": wv, "hbm_bytes_per_launch": int((2 * fv + wv) * 1024)})
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)

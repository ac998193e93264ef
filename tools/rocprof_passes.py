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

# Overlapped second kNN pass (k_knn<256, true> on its own stream beside the first solve): its dispatch lasts from the start of
# the first solve to a few microseconds after it, so the per-scan chain is the meaningful figure.
rows = db.execute("select name, start, end from kernels where name like '%k_knn%' or name like '%k_lm_solve%' or name like '%k_rebuild_alloc%' order by start").fetchall()
if any("true" in r[0] for r in rows):
    scans, cur = [], None
    for name, a, b in rows:
        if "k_knn" in name and "true" not in name:
            if cur and len(cur) == 5:
                scans.append(cur)
            cur = {"k0": (a, b)}
        elif cur is not None:
            if "k_knn" in name:
                cur["k1"] = (a, b)
            elif "k_rebuild_alloc" in name:
                cur["al"] = (a, b)
            elif "l0" not in cur:
                cur["l0"] = (a, b)
            else:
                cur["l1"] = (a, b)
    scans = scans[len(scans) // 10: len(scans) // 5] if len(scans) >= 500 else scans[-150:]
    scans = [c for c in scans if c["l1"][1] - c["k0"][0] < 300e3]
    if scans:
        avg = lambda f: sum(f(c) for c in scans) / len(scans) / 1e3
        print("overlapped scans (n=%d): first pass %.2f us, first solve %.2f, second pass ends %.2f after the first solve, k_rebuild_alloc %.2f, "
              "finalising solve starts %.2f after the first and lasts %.2f; first pass start -> finalising solve end %.2f us" %
              (len(scans), avg(lambda c: c["k0"][1] - c["k0"][0]), avg(lambda c: c["l0"][1] - c["l0"][0]), avg(lambda c: c["k1"][1] - c["l0"][1]),
               avg(lambda c: c["al"][1] - c["al"][0]), avg(lambda c: c["l1"][0] - c["l0"][1]), avg(lambda c: c["l1"][1] - c["l1"][0]),
               avg(lambda c: c["l1"][1] - c["k0"][0])))

# Chain mode (round 5): k_knn<256, false, true> (first pass) | k_ov_gate | k_knn<256, true, false> (+ COUNT / PAD) | k_rebuild_alloc |
# k_rebuild_fin on one stream, the two k_lm_solve launches (solving workgroups only) on the other; a solve launch is RESIDENT while
# the pass before it still runs, so its duration is not its work: the phases are measured between the ends of the links.
rows = db.execute("select name, start, end from kernels where name like '%k_knn%' or name like '%k_lm_solve%' or name like '%k_rebuild_fin%' order by start").fetchall()
if any("false, true" in r[0] for r in rows):
    k0 = [r for r in rows if "k_knn" in r[0] and "false, true" in r[0]]
    k1 = [r for r in rows if "k_knn" in r[0] and "true, false" in r[0]]
    lm = [r for r in rows if "k_lm_solve" in r[0]]
    fin = [r for r in rows if "k_rebuild_fin" in r[0]]
    scans = []
    for i in range(len(k0) - 1):
        a, nxt = k0[i], k0[i + 1]
        b1 = [e for e in k1 if a[1] <= e[1] < nxt[1]]
        fn = [e for e in fin if a[1] <= e[1] < nxt[2]]
        # the scan's two solves: those that END after this first pass has ended and before the next first pass has (the first
        # solve of the next scan is resident by then but ends later)
        sol = [e for e in lm if a[2] <= e[2] <= nxt[2]]
        if len(b1) != 1 or len(fn) != 1 or len(sol) != 2:
            continue
        s0, s1 = sol
        end = max(s1[2], fn[0][2])
        scans.append((a[2] - a[1], s0[2] - a[2], b1[0][2] - s0[2], s1[2] - b1[0][2], end - s1[2], end - a[1], nxt[1] - a[1], s0[1] - a[1]))
    scans = [c for c in scans if c[6] < 300e3]
    scans = scans[len(scans) // 10: len(scans) // 5] if len(scans) >= 500 else scans[-150:]
    if scans:
        import statistics as st
        col = lambda j: st.median(c[j] for c in scans) / 1e3
        print("chain-mode scans (n=%d, medians): first pass %.2f us; first solve ends %.2f after it (its launch starts %.2f after the pass's); second pass "
              "ends %.2f after the first solve; finalising solve ends %.2f after the second pass; APPEND launch ends %.2f after that; first pass start -> "
              "end of the scan %.2f us; period %.2f us" % (len(scans), col(0), col(1), col(7), col(2), col(3), col(4), col(5), col(6)))

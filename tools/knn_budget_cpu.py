#!/usr/bin/env python3
"""CPU-side budget of the first kNN pass (no GPU): for the queries and the 20-frame window of the headline stream (scan 30 of
bench.py's synthetic HDL-64 stream, taken from the oracle's run) — how many map points a query has to look at under different cell
layouts and pruning rules, and how the rounds of a wave that holds several queries add up.  Everything k_knn8's design (kernels_knn8.h)
quotes comes from here.  usage: python tools/knn_budget_cpu.py [scan=30]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from liodom_amd import synth
from oracle import oracle as orc
from scipy.spatial import cKDTree

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
H, W, R, epr, P = 64, 1800, 8, 10, 20
cfg = synth.make_cfg(H, W, 0)
po = orc.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, knn_mode=1)
od = orc.Odometer(po)
frames = []
for k in range(K + 1):
    x, _ = synth.scan(cfg, 0, k)
    e = orc.extract(po, x, H, W)
    if k == K:
        win = od.window().copy()
    od.step(e["edges"])
    frames.append(od.window()[-len(e["edges"]):, :3].copy())
q = od.last_queries(0)[:, :3].astype(np.float32)
q = q[np.isfinite(q).all(axis=1)]
win = win[:, :3].astype(np.float32)
print("scan %d: %d queries, %d window points" % (K, len(q), len(win)))
tree = cKDTree(win.astype(np.float64))
d5 = tree.query(q.astype(np.float64), k=5)[0][:, 4]
print("fifth-nearest distance (m): percentiles 10/25/50/75/90/99 = %s; inside the 1.0 gate: %.1f %%" % (np.round(np.percentile(d5, [10, 25, 50, 75, 90, 99]), 3).tolist(), 100 * (d5 < 1).mean()))


def keys(p, cs=1.0):
    c = np.floor(p / cs).astype(np.int64)
    return (c[:, 0] + 2 ** 20) + ((c[:, 1] + 2 ** 20) << 21) + ((c[:, 2] + 2 ** 20) << 42)


def key1(c):
    return int((c[0] + 2 ** 20) + ((c[1] + 2 ** 20) << 21) + ((c[2] + 2 ** 20) << 42))


def build(cs):
    k = keys(win, cs)
    order = np.argsort(k, kind="stable")
    u, start, cnt = np.unique(k[order], return_index=True, return_counts=True)
    return dict(zip(u.tolist(), zip(start.tolist(), cnt.tolist()))), win[order]


def sqd(p, m):
    dx = p[0] - m[:, 0]; dy = p[1] - m[:, 1]; dz = p[2] - m[:, 2]
    r = dx * dx; r = r + dy * dy; r = r + dz * dz
    return r.astype(np.float32)


cells, pts = build(1.0)
print("1 m cells: %d occupied, %.1f points each on average, %d at most" % (len(cells), np.mean([c[1] for c in cells.values()]), max(c[1] for c in cells.values())))
own, all27, ideal = [], [], []
for i, p in enumerate(q):
    c = np.floor(p).astype(np.int64)
    lst = []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                cc = c + np.array([dx, dy, dz])
                sc = cells.get(key1(cc))
                if sc is None:
                    continue
                lo = cc.astype(np.float32); hi = lo + 1
                e = np.maximum(np.maximum(lo - p, p - hi), 0)
                lst.append((float((e * e).sum()), sc[0], sc[1], dx == 0 and dy == 0 and dz == 0))
    own.append(sum(n for _, _, n, o in lst if o)); all27.append(sum(n for _, _, n, _ in lst))
    lst.sort()
    best = np.full(5, 1.0, np.float32); n = 0
    for lb, s, cn, _ in lst:                      # cells in the order of their box distance, pruned with the running fifth distance
        if lb > best[4]:
            break
        best = np.sort(np.concatenate([best, sqd(p, pts[s:s + cn])]))[:5]; n += cn
    ideal.append(n)
own, all27, ideal = map(np.array, (own, all27, ideal))
for name, a in (("own 1 m cell", own), ("all 27 cells", all27), ("cells within the running fifth distance (ideal pruning)", ideal)):
    print("candidates per query, %-58s mean %6.1f  median %4.0f  p90 %4.0f  max %5d" % (name + ":", a.mean(), np.median(a), np.percentile(a, 90), a.max()))

# ---- k_knn8's scheme: 8 lanes per query, Best3 per lane, own cell -> ladder bound -> the neighbour cells within it ----
G = 8
thr = [0.36, 0.09, 0.0225, 0.0036]
visited, nbcells, fails = [], [], 0
nb_lists = []                 # per query: the populations of the neighbour cells it streams
for p in q:
    c = np.floor(p).astype(np.int64)
    m = np.full((G, 4), np.inf, np.float32)

    def stream(s, n):
        dd = sqd(p, pts[s:s + n])
        for i in range(n):
            l = i % G
            v = np.sort(np.append(m[l], dd[i]))
            m[l] = v[:4]
    sc = cells.get(key1(c)); nown = 0
    if sc:
        stream(*sc); nown = sc[1]
    kept = np.sort(m[:, :3].ravel())
    B = np.float32(1.0)
    for t in thr:
        if (kept <= np.float32(t)).sum() >= 5:
            B = np.float32(t)
    v, nb = nown, 0
    mine = []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dx == 0 and dy == 0 and dz == 0:
                    continue
                cc = c + np.array([dx, dy, dz]); lo = cc.astype(np.float32); hi = lo + 1
                e = np.maximum(np.maximum(lo - p, p - hi), 0).astype(np.float32)
                if np.float32((e * e).sum() * np.float32(1 - 1e-5)) > B:
                    continue
                sc2 = cells.get(key1(cc))
                if sc2:
                    stream(*sc2); v += sc2[1]; nb += 1; mine.append(sc2[1])
    ent = np.sort(m[:, :3].ravel()); g4 = ent[4]; m4 = m[:, 3].min()
    ok = (g4 < 1.0 and all(ent[i] < ent[i + 1] for i in range(5)) and g4 < m4) or (g4 >= 1.0 and m4 >= 1.0)
    fails += 0 if ok else 1
    visited.append(v); nbcells.append(nb); nb_lists.append(mine)
visited = np.array(visited)
print("k_knn8 (8 lanes, Best3, ladder bound after the own cell, lazy neighbours): candidates per query mean %.1f median %.0f p90 %.0f; neighbour cells streamed %.2f per query; results the fast path cannot certify: %d of %d (%.2f %%)" % (
    visited.mean(), np.median(visited), np.percentile(visited, 90), np.mean(nbcells), fails, len(q), 100.0 * fails / len(q)))


def rounds(a, per_wg):
    it = np.ceil(a / G); pad = (-len(it)) % per_wg
    it = np.concatenate([it, np.zeros(pad)])
    tot_u = it.reshape(-1, 8).max(axis=1).sum()
    srt = np.concatenate([np.sort(w) for w in it.reshape(-1, per_wg)])
    return tot_u / (len(it) / 8), srt.reshape(-1, 8).max(axis=1).sum() / (len(it) / 8)
for per_wg in (32, 64):
    u, s_ = rounds(own, per_wg)
    print("rounds of the own cell per wave of 8 queries (a wave lasts as long as its most populous cell): edge order %.1f; workgroup of %d queries sorted by cell population %.1f; all equal %.1f" % (u, per_wg, s_, own.mean() / G))

# ---- neighbour cells: rounds per wave under different walks (the groups of a wave hold different lists) ----
def nb_rounds(per_wg=32):
    n = len(nb_lists); pad = (-n) % per_wg
    lists = nb_lists + [[]] * pad
    order_own = np.concatenate([own, np.zeros(pad)])
    res = {}
    def walk(perm, step):
        cell_by_cell = one_stream = 0.0
        for w in range(0, len(perm), 8):
            grp = [lists[i] for i in perm[w:w + 8]]
            depth = max(len(g) for g in grp)
            cell_by_cell += sum(max((np.ceil(g[k] / step) if k < len(g) else 0) for g in grp) for k in range(depth))
            one_stream += max(sum(np.ceil(c / step) for c in g) for g in grp)
        return cell_by_cell / (len(perm) / 8), one_stream / (len(perm) / 8)
    perm_own = np.concatenate([w0 + np.argsort(order_own[w0:w0 + per_wg], kind="stable") for w0 in range(0, len(lists), per_wg)])
    for step in (8, 16):
        work = np.array([sum(np.ceil(c / step) for c in g) for g in lists])
        perm_work = np.concatenate([w0 + np.argsort(work[w0:w0 + per_wg], kind="stable") for w0 in range(0, len(lists), per_wg)])
        a, b = walk(perm_own, step)
        c_, d = walk(perm_work, step)
        print("neighbour cells, steps of %2d candidates per group, per wave of 8 queries: mean over queries %.1f; cell by cell %.1f; one stream per group %.1f; "
              "workgroup of %d re-dealt by neighbour work: cell by cell %.1f, one stream %.1f" % (step, work.mean(), a, b, per_wg, c_, d))
nb_rounds(32)
nb_rounds(64)

# ---- sub-cell ordering (the round-5 verdict's lever (a)): the points of a populous cell sorted by 0.5 m octant, a query walks its
# ---- own octant, forms the ladder bound, then only the octants (own cell and neighbours) whose box lies within it ----
def octant_model(heavy):
    c0 = np.floor(win)
    ob = ((win - c0) >= 0.5).astype(np.int64)
    oc = ob[:, 0] + 2 * ob[:, 1] + 4 * ob[:, 2]
    kk = keys(win)
    order = np.lexsort((oc, kk))
    ks, ocs, pp = kk[order], oc[order], win[order]
    u, st_, cn = np.unique(ks, return_index=True, return_counts=True)
    cl = {a: (b, c, np.bincount(ocs[b:b + c], minlength=8)) for a, b, c in zip(u.tolist(), st_.tolist(), cn.tolist())}
    tot, r8 = [], []
    for p in q:
        c = np.floor(p).astype(np.int64)
        m = np.full((G, 4), np.inf, np.float32)
        seen = rounds8 = 0
        def stream(s, n):
            nonlocal seen, rounds8
            if n <= 0:
                return
            dd = sqd(p, pp[s:s + n])
            for i in range(n):
                m[i % G] = np.sort(np.append(m[i % G], dd[i]))[:4]
            seen += n; rounds8 += -(-n // 8)
        pending = []
        own_oct = int(p[0] - c[0] >= 0.5) + 2 * int(p[1] - c[1] >= 0.5) + 4 * int(p[2] - c[2] >= 0.5)
        sc = cl.get(key1(c))
        if sc:
            s, n, sub = sc
            if n > heavy:
                offs = np.concatenate([[0], np.cumsum(sub)])
                stream(s + offs[own_oct], sub[own_oct])
                pending += [(c, o, s + offs[o], sub[o]) for o in range(8) if o != own_oct and sub[o]]
            else:
                stream(s, n)
        kept = np.sort(m[:, :3].ravel()); B = np.float32(1.0)
        for t in thr:
            if (kept <= np.float32(t)).sum() >= 5:
                B = np.float32(t)
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    if dx == 0 and dy == 0 and dz == 0:
                        continue
                    cc = c + np.array([dx, dy, dz]); lo = cc.astype(np.float32); hi = lo + 1
                    e = np.maximum(np.maximum(lo - p, p - hi), 0).astype(np.float32)
                    if np.float32((e * e).sum()) > B:
                        continue
                    sc2 = cl.get(key1(cc))
                    if not sc2:
                        continue
                    s, n, sub = sc2
                    if n > heavy:
                        offs = np.concatenate([[0], np.cumsum(sub)])
                        pending += [(cc, o, s + offs[o], sub[o]) for o in range(8) if sub[o]]
                    else:
                        stream(s, n)
        for cc, o, s, n in pending:
            lo = cc.astype(np.float32) + 0.5 * np.array([o & 1, (o >> 1) & 1, (o >> 2) & 1], np.float32); hi = lo + 0.5
            e = np.maximum(np.maximum(lo - p, p - hi), 0).astype(np.float32)
            if np.float32((e * e).sum()) <= B:
                stream(s, n)
        tot.append(seen); r8.append(rounds8)
    tot, r8 = np.array(tot), np.array(r8)
    pad = (-len(r8)) % 8
    per_wave = np.concatenate([r8, np.zeros(pad)]).reshape(-1, 8).max(axis=1).mean()
    return tot.mean(), np.median(tot), np.percentile(tot, 90), r8.mean(), per_wave
if os.environ.get("KNN_BUDGET_OCTANTS", "1") != "0":
    n_heavy = sum(1 for v_ in cells.values() if v_[1] > 32)
    print("sub-cell ordering: %d of %d cells hold more than 32 points (%d of the %d window points)" % (n_heavy, len(cells), sum(v_[1] for v_ in cells.values() if v_[1] > 32), len(win)))
    for heavy, name in ((10 ** 9, "no octants (k_knn8 as built)"), (64, "cells > 64 points sorted by octant"), (32, "cells > 32 points sorted by octant"), (16, "cells > 16 points sorted by octant")):
        a = octant_model(heavy)
        print("   %-38s candidates per query mean %5.1f median %3.0f p90 %3.0f; rounds of 8 per query %.1f, per wave of 8 queries (edge order) %.1f" % ((name + ":",) + a))

# ---- k_hash_append: room per cell ----
def sim(slack_fn, newroom, period=4):
    fails = tries = 0
    for f0 in range(max(P, 22), K - period, period):
        k = keys(np.concatenate(frames[f0 - P + 1:f0 + 1]))
        u, c = np.unique(k, return_counts=True)
        room = {kk: slack_fn(v) for kk, v in zip(u.tolist(), c.tolist())}
        for f in range(f0 + 1, f0 + period):
            uu, cc = np.unique(keys(frames[f]), return_counts=True)
            bad = False
            for a, b in zip(uu.tolist(), cc.tolist()):
                room[a] = room.get(a, newroom) - b
                bad = bad or room[a] < 0
            tries += 1
            if bad:
                fails += 1
                break
    return fails, tries
inc = np.concatenate([np.unique(keys(f), return_counts=True)[1] for f in frames[-8:]])
print("k_hash_append: a frame puts %.1f points into a cell it touches (p99 %d, max %d), %d cells per frame" % (inc.mean(), np.percentile(inc, 99), inc.max(), len(inc) // 8))
for name, fn, nr in (("a quarter of the population, >= 8; new cells 16", lambda c: max(8, c // 4), 16), ("the population again, >= 32; new cells 96", lambda c: max(32, c), 96)):
    f, t = sim(fn, nr)
    print("   room = %-50s appends that run out of room: %d of %d" % (name + ":", f, t))

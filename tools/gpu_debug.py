"""Diagnostic run on the GPU box: prints detailed GPU-vs-oracle comparisons instead of asserting.
Usage: python tools/gpu_debug.py [section ...]"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liodom_amd as la  # noqa: E402
from liodom_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def mk(H, W, lt=0, R=8, epr=10, P=5, S=1, debug=0, **kw):
    po = orc.make_params(lidar_type=lt, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, knn_mode=1)
    g = la.Liodom(la.make_params(lidar_type=lt, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
                  la.make_config(n_streams=S, max_points=H * W, max_width=W, debug_buffers=debug, **kw))
    return po, g


def sec_extract():
    for (H, W, lt, R, epr) in [(16, 900, 0, 6, 10), (64, 1800, 0, 8, 10), (128, 2048, 1, 8, 10)]:
        cfg = synth.make_cfg(H, W, lt)
        po, g = mk(H, W, lt, R, epr, debug=1)
        x, _ = synth.scan(cfg, 0, 0)
        o = orc.extract(po, x, H, W, want_curv=True)
        t = time.time()
        e = g.extract_edges(x, H, W)
        dt = time.time() - t
        print("extract %dx%d: gpu %d edges, oracle %d edges, %.2f ms (incl. copies)" % (H, W, len(e["ring"]), len(o["ring"]), dt * 1e3))
        cg, offs = g.curvature()
        offs_o, _ = orc.split(po, x, H, W)
        print("  ring offsets equal:", np.array_equal(offs, offs_o))
        if np.array_equal(offs, offs_o):
            co = o["curv"][:len(cg)]
            m = ~(np.isnan(cg) & np.isnan(co))
            bad = (cg[m].view(np.uint64) != co[m].view(np.uint64))
            print("  curvature mismatches: %d of %d" % (bad.sum(), m.sum()))
            if bad.sum():
                i = np.nonzero(m)[0][np.nonzero(bad)[0][:5]]
                print("   first:", i, cg[i], co[i])
        else:
            print("  gpu ring sizes", np.diff(offs)[:20], "\n  orc ring sizes", np.diff(offs_o)[:20])
        n = min(len(e["ring"]), len(o["ring"]))
        same = (e["ring"][:n] == o["ring"][:n]) & (e["idx_in_ring"][:n] == o["idx_in_ring"][:n]) & (e["src"][:n] == o["src"][:n])
        print("  edges identical: %s (first diff at %s)" % (bool(same.all() and len(e["ring"]) == len(o["ring"])), np.nonzero(~same)[0][:3]))
        if not same.all():
            i = np.nonzero(~same)[0][0]
            sl = slice(max(0, i - 3), i + 6)
            print("   gpu", list(zip(e["ring"][sl], e["idx_in_ring"][sl], e["src"][sl])))
            print("   orc", list(zip(o["ring"][sl], o["idx_in_ring"][sl], o["src"][sl])))
        g.close()


def sec_odom(H=16, W=900, lt=0, R=6, epr=10, P=5, K=12):
    cfg = synth.make_cfg(H, W, lt)
    po, g = mk(H, W, lt, R, epr, P)
    od = orc.Odometer(po)
    for k in range(K):
        x, gt = synth.scan(cfg, 0, k)
        o = orc.extract(po, x, H, W)
        pose_o, io = od.step(o["edges"])
        pose_g, ig = g.process_scan(x, H, W)
        dt = np.linalg.norm(pose_g[4:] - pose_o[4:])
        dq = min(np.linalg.norm(pose_g[:4] - pose_o[:4]), np.linalg.norm(pose_g[:4] + pose_o[:4]))
        line = "scan %2d E %d/%d M %d/%d match %s/%s it %s/%s term %s/%s dt %.2e dq %.2e" % (
            k, ig.n_edges, io.n_edges, ig.map_points, io.map_points, list(ig.matches), list(io.matches),
            [ig.lm[0].iterations, ig.lm[1].iterations], [io.lm[0].iterations, io.lm[1].iterations],
            [ig.lm[0].termination, ig.lm[1].termination], [io.lm[0].termination, io.lm[1].termination], dt, dq)
        if k > 0:
            for it in (0, 1):
                vo, ao, bo = od.last_corr(it)
                vg, ag, bg = g.correspondences(it)
                nd = int((vo != vg).sum()) + int(((ao != ag) | (bo != bg))[(vo == 1) & (vg == 1)].sum())
                line += " corrdiff%d=%d" % (it, nd)
            line += " cost %.6g/%.6g" % (ig.lm[1].final_cost, io.lm[1].final_cost)
        wo = od.window()
        wg, nf = g.window()
        line += " win %s/%s maxdiff %.2e" % (wg.shape[0], wo.shape[0], np.abs(wg - wo).max() if wg.shape == wo.shape and len(wo) else -1)
        print(line)
        if ig.status:
            print("  STATUS", ig.status)
    g.close()


def sec_timing(H=64, W=1800, lt=0, R=8, epr=10, P=20, K=60, S=1):
    cfg = synth.make_cfg(H, W, lt)
    po, g = mk(H, W, lt, R, epr, P, S=S, pose_log_capacity=4 * K)
    print("device:", g.device_info())
    g.alloc_resident(K)
    for s in range(S):
        for k in range(K):
            g.upload_scan(s, k, synth.scan(cfg, s, k)[0])
    for mode in ("sync", "async"):
        g.reset()
        for k in range(10):
            g.process_resident(k, H * W, H, W, readback=True)
        g.sync()
        t = time.time()
        for k in range(10, K):
            g.process_resident(k, H * W, H, W, readback=(mode == "sync"))
        g.sync()
        dt = time.time() - t
        print("S=%d %s: %.1f us/scan-step, %.0f scans/s aggregate" % (S, mode, dt / (K - 10) * 1e6, S * (K - 10) / dt))
    g.reset()
    g.set_profiling(True)
    for k in range(K):
        g.process_resident(k, H * W, H, W, readback=False)
    st = g.kernel_stats()
    g.set_profiling(False)
    tot = sum(v[1] for v in st.values())
    for name, (n, ms) in st.items():
        if n:
            print("  %-16s %5d launches  avg %8.2f us  total %7.2f ms (%.0f%%)" % (name, n, ms / n * 1e3, ms, 100 * ms / tot))
    _, infos = g.pose_log(0, K - 1, 1)
    print("  last scan: E=%d M=%d matches=%s" % (infos[0].n_edges, infos[0].map_points, list(infos[0].matches)))
    g.close()


SECTIONS = {
    "extract": sec_extract,
    "odom": sec_odom,
    "odom64": lambda: sec_odom(64, 1800, 0, 8, 10, 20, 14),
    "timing": sec_timing,
    "timing_batch": lambda: [sec_timing(S=s, K=40) for s in (8, 64)],
}

if __name__ == "__main__":
    names = sys.argv[1:] or ["extract", "odom", "odom64", "timing"]
    for n in names:
        print("=" * 20, n)
        try:
            SECTIONS[n]()
        except Exception:
            traceback.print_exc()
        sys.stdout.flush()

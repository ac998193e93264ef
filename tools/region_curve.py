#!/usr/bin/env python3
"""T(K): the length of bench.py's timed region as a function of its K (headline shape, chain mode, one stream) — what a region
costs beyond K periods (the driver times K = 20).  usage: python tools/region_curve.py [repeats=9]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
try:
    os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[1]})
except Exception:
    pass
import liodom_amd as la
from liodom_amd import synth
la.load()
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 9
H, W, R, epr, P = 64, 1800, 8, 10, 20
N = H * W
F, Wm = P, 5
KS = [1, 2, 3, 5, 10, 20, 40, 100]
total = F + Wm + max(KS)
cfg = synth.make_cfg(H, W, 0)
scans = [synth.scan(cfg, 0, k)[0] for k in range(total + 1)]
g = la.Liodom(la.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P), la.make_config(n_streams=1, max_points=N, max_width=W, pose_log_capacity=total + 8))
g.alloc_resident(total + 1)
for k in range(total + 1):
    g.upload_scan(0, k, scans[k])
g.sync()
time.sleep(0.3)
res = {}
for K in KS:
    t = []
    for r in range(REP + 1):
        g.reset()
        g.replay_resident(0, F, N, H, W, depth=1, ahead=False)
        g.replay_resident(F, Wm, N, H, W, depth=1, ahead=True)
        g.sync()
        t0 = time.perf_counter()
        g.replay_resident(F + Wm, K, N, H, W, depth=1, ahead=True)
        t1 = time.perf_counter()
        g.sync()
        t2 = time.perf_counter()
        t.append((t2 - t0, t1 - t0))
    t = sorted(t[1:])
    res[K] = t[len(t) // 2]
    print("K = %3d: region %8.1f us (replay call %8.1f, final sync %5.1f) = %6.1f us per scan" % (K, res[K][0] * 1e6, res[K][1] * 1e6, (res[K][0] - res[K][1]) * 1e6, res[K][0] * 1e6 / K))
b = (res[100][0] - res[20][0]) / 80
print("period from K = 20 -> 100: %.2f us; fixed cost of a region at K = 20: %.1f us; first scan alone %.1f us" % (b * 1e6, (res[20][0] - 20 * b) * 1e6, res[1][0] * 1e6))
import ctypes as C
g.L.liodom_debug_replay_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
acc = []
for r in range(REP):
    g.reset()
    g.replay_resident(0, F, N, H, W, depth=1, ahead=False)
    g.replay_resident(F, Wm, N, H, W, depth=1, ahead=True)
    g.sync()
    g.replay_resident(F + Wm, 60, N, H, W, depth=1, ahead=True)
    buf = (C.c_double * 64)()
    n = g.L.liodom_debug_replay_stamps(g.h, buf, 64)
    acc.append(np.array(buf[:n]))
    g.sync()
a = np.median(np.stack(acc), axis=0)
print("pose arrival times in a K = 60 region (us since the call, median of %d): first %.1f; intervals:" % (REP, a[0]))
print("  " + " ".join("%.1f" % x for x in np.diff(a)))
print(g.modes().get("chain"), g.modes().get("speculate"), "spec_early", g.modes().get("spec_early"), "unconfirmed", g.modes().get("spec_unconfirmed"))
g.close()

"""Fixed cost of bench.py's timed region: the region (reset, pre-fill, warm-up, sync | K steps, sync) for several K; the slope is the
steady-state period, the intercept what the region pays once (pipeline fill, drain, synchronisation).  usage: python tools/region_fixed_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("REGION_LIB"):          # a variant built by tools/variant_build.sh instead of the product library
    import liodom_amd.api as _api
    _api._LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "variants", "lib%s.so" % os.environ["REGION_LIB"])
    _api.is_stale = lambda: False
import liodom_amd as la
from liodom_amd import synth
H, W, lt, R, epr, P = 64, 1800, 0, 8, 10, 20
N = H * W
F, Wm = P + 10, 5
Ks = [10, 20, 40, 80, 200]
total = F + Wm + max(Ks) + 1
cfg = synth.make_cfg(H, W, lt)
g = la.Liodom(la.make_params(lidar_type=lt, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P), la.make_config(n_streams=1, max_points=N, max_width=W, pose_log_capacity=total + 8))
g.alloc_resident(total)
for k in range(total):
    g.upload_scan(0, k, synth.scan(cfg, 0, k)[0])
g.sync()
res = {}
for rep in range(7):
    for K in Ks:
        g.reset()
        g.replay_resident(0, F, N, H, W, depth=1, ahead=True)
        g.replay_resident(F, Wm, N, H, W, depth=1, ahead=True)
        g.sync()
        t0 = time.perf_counter()
        g.replay_resident(F + Wm, K, N, H, W, depth=1, ahead=True)
        t1 = time.perf_counter()
        g.sync()
        t2 = time.perf_counter()
        if rep:
            res.setdefault(K, []).append(((t1 - t0) * 1e6, (t2 - t1) * 1e6))
med = {K: (np.median([a for a, b in v]), np.median([b for a, b in v])) for K, v in res.items()}
for K in Ks:
    a, b = med[K]
    print("K = %3d: replay call %8.1f us + sync %6.1f us = %8.1f us -> %8.1f scans/s" % (K, a, b, a + b, K / (a + b) * 1e6))
x = np.array(Ks, dtype=float)
y = np.array([med[K][0] + med[K][1] for K in Ks])
slope, icpt = np.polyfit(x, y, 1)
print("fit: %.2f us per scan + %.1f us per region" % (slope, icpt))
g.close()

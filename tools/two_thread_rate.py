"""Rate of the two-thread binding (liodom_host_two_thread_replay) on a BASELINE workload; environment switches are read at
handle creation.  usage: python tools/two_thread_rate.py [hdl64|vlp16|ouster128] [scans] [repeats]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liodom_amd as la
from liodom_amd import synth
WL = {"hdl64": (64, 1800, 0, 8, 10, 20), "vlp16": (16, 1800, 0, 8, 20, 10), "ouster128": (128, 2048, 1, 8, 10, 30)}
name = sys.argv[1] if len(sys.argv) > 1 else "hdl64"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 240
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
H, W, lt, R, epr, P = WL[name]
N = H * W
cfg = synth.make_cfg(H, W, lt)
F = P + 10
scans = np.stack([synth.scan(cfg, 0, k)[0] for k in range(F + K)])
g = la.Liodom(la.make_params(lidar_type=lt, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
              la.make_config(n_streams=1, max_points=N, max_width=W, pose_log_capacity=F + K + 8))
for depth in (1, 0):
    for fetch in (True, False):
        rates = []
        for r in range(reps):
            g.reset()
            poses, secs, tot = g.two_thread_replay(scans, N, H, W, timed_from=F, fetch_edges=fetch, depth=depth)
            rates.append(K / secs)
        rates.sort()
        print("%s two-thread depth %d fetch_edges %d: median %.1f scans/s (min %.1f max %.1f)  env %s" % (
            name, depth, fetch, rates[len(rates) // 2], rates[0], rates[-1],
            {k: v for k, v in os.environ.items() if k.startswith("LIODOM_")}), flush=True)
g.close()

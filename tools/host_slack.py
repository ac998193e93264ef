"""How much host time per scan the per-scan synchronous replay can absorb before the GPU starves: the headline loop of
bench.py with a busy-wait of X us between two calls.  Usage: python tools/host_slack.py [X ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liodom_amd as la  # noqa: E402
from liodom_amd import synth  # noqa: E402

H, W, R, epr, P = 64, 1800, 8, 10, 20
K, F = 200, 40
N = H * W
cfg = synth.make_cfg(H, W, 0)
scans = [synth.scan(cfg, 0, k)[0] for k in range(F + K)]
g = la.Liodom(la.make_params(lidar_type=0, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
              la.make_config(n_streams=1, max_points=N, max_width=W, pose_log_capacity=F + K + 8))
g.alloc_resident(F + K)
for k in range(F + K):
    g.upload_scan(0, k, scans[k])
g.sync()
time.sleep(0.5)
for X in [float(a) for a in (sys.argv[1:] or ["0", "2", "5", "10", "15", "20"])]:
    g.reset()
    for k in range(F):
        g.process_resident(k, N, H, W, readback=True, next_slot=k + 1)
    g.sync()
    t0 = time.perf_counter()
    for k in range(F, F + K):
        g.process_resident(k, N, H, W, readback=True, next_slot=(k + 1 if k + 1 < F + K else -1))
        if X > 0:
            t1 = time.perf_counter() + X * 1e-6
            while time.perf_counter() < t1:
                pass
    g.sync()
    dt = time.perf_counter() - t0
    print("host delay %5.1f us per scan: %8.1f scans/s (%.2f us per scan)" % (X, K / dt, dt / K * 1e6))
# the same loop in C (liodom_replay_resident), alternating with the Python loop
def py_loop():
    g.reset()
    for k in range(F):
        g.process_resident(k, N, H, W, readback=True, next_slot=k + 1)
    g.sync()
    t0 = time.perf_counter()
    for k in range(F, F + K):
        g.process_resident(k, N, H, W, readback=True, next_slot=(k + 1 if k + 1 < F + K else -1))
    g.sync()
    return time.perf_counter() - t0


def c_loop():
    g.reset()
    g.replay_resident(0, F, N, H, W, ahead=True)
    g.sync()
    t0 = time.perf_counter()
    g.replay_resident(F, K, N, H, W)
    g.sync()
    return time.perf_counter() - t0


for rep in range(4):
    a, b = py_loop(), c_loop()
    print("python loop %8.1f scans/s (%.2f us)   C loop %8.1f scans/s (%.2f us)" % (K / a, a / K * 1e6, K / b, b / K * 1e6))
g.close()

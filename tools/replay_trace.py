"""Resident depth-1 replay of a BASELINE workload (for kernel traces).  usage: python tools/replay_trace.py [hdl64|vlp16|ouster128] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("PIN_CPUS"):          # e.g. PIN_CPUS=0-63 or 8: where the enqueueing thread runs (NUMA distance to the GPU shows in replay_enqueue_us)
    cpus = set()
    for part in os.environ["PIN_CPUS"].split(","):
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    os.sched_setaffinity(0, cpus)
import liodom_amd as la
from liodom_amd import synth
WL = {"hdl64": (64, 1800, 0, 8, 10, 20), "vlp16": (16, 1800, 0, 8, 20, 10), "ouster128": (128, 2048, 1, 8, 10, 30)}
name = sys.argv[1] if len(sys.argv) > 1 else "hdl64"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
H, W, lt, R, epr, P = WL[name]
F = P + 10
cfg = synth.make_cfg(H, W, lt)
g = la.Liodom(la.make_params(lidar_type=lt, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
              la.make_config(n_streams=1, max_points=H * W, max_width=W, pose_log_capacity=F + K + 8))
g.alloc_resident(F + K)
for k in range(F + K):
    g.upload_scan(0, k, synth.scan(cfg, 0, k)[0])
g.replay_resident(0, F, H * W, H, W, depth=1, ahead=True)
t = time.perf_counter()
g.replay_resident(F, K, H * W, H, W, depth=1)
dt = time.perf_counter() - t
m = g.modes()
print("%s: %.1f scans/s (%.2f us/scan)  chain=%s knn_overlap=%s speculate=%s; host per scan: enqueue %s us, waiting for the previous pose %s us" % (
    name, K / dt, dt / K * 1e6, m.get("chain"), m.get("knn_overlap"), m.get("speculate"), m.get("replay_enqueue_us"), m.get("replay_wait_us")))
g.close()

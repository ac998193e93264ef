"""Determinism / race stress on the headline shape: the same 240 scans replayed in every mode (consumer loop depth 1 and 0,
asynchronous, strictly serial, per-call pipelined), several times each; every pose log must equal the first bit for bit
and no status bit may be raised.  Usage: python tools/stress_modes.py [repeats [hdl64|ouster128|vlp16]]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liodom_amd as la  # noqa: E402
from liodom_amd import synth  # noqa: E402

SHAPES = {"hdl64": (64, 1800, 0, 8, 10, 20, 240), "ouster128": (128, 2048, 1, 8, 10, 30, 120), "vlp16": (16, 1800, 0, 8, 20, 10, 240)}
H, W, LT, R, epr, P, K = SHAPES[sys.argv[2] if len(sys.argv) > 2 else "hdl64"]
N = H * W
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = synth.make_cfg(H, W, LT)
scans = [synth.scan(cfg, 3, k)[0] for k in range(K)]
g = la.Liodom(la.make_params(lidar_type=LT, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
              la.make_config(n_streams=1, max_points=N, max_width=W, pose_log_capacity=K + 8))
g.alloc_resident(K)
for k in range(K):
    g.upload_scan(0, k, scans[k])


def run(mode):
    g.reset()
    if mode == "depth1":
        g.replay_resident(0, K, N, H, W, depth=1)
    elif mode == "depth0":
        g.replay_resident(0, K, N, H, W, depth=0)
    elif mode == "async":
        for k in range(K):
            g.process_resident(k, N, H, W, readback=False, next_slot=(k + 1 if k + 1 < K else -1))
    elif mode == "serial":
        for k in range(K):
            g.process_resident(k, N, H, W, readback=True)
    elif mode == "percall":
        for k in range(K):
            g.process_resident(k, N, H, W, readback=True, next_slot=(k + 1 if k + 1 < K else -1))
    g.sync()
    poses, infos = g.pose_log(0, 0, K)
    st = 0
    for i in infos:
        st |= int(i.status)
    return poses.copy(), st


ref = None
t0 = time.time()
for rep in range(reps):
    for mode in ("depth1", "depth0", "async", "serial", "percall"):
        poses, st = run(mode)
        if ref is None:
            ref = poses
        same = np.array_equal(ref.view(np.uint64), poses.view(np.uint64))
        print("repeat %d %-8s status 0x%x identical %s" % (rep, mode, st, same))
        if st or not same:
            raise SystemExit("FAILED")
print("all %d replays identical, %.1f s" % (reps * 5, time.time() - t0))
g.close()

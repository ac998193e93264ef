import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import liodom_amd as la
from liodom_amd import synth
H, W, lt, R, epr, P = 64, 1800, 0, 8, 10, 20
N = H * W
F, Wm, K = P + 10, 5, 10
total = F + Wm + K + 1
cfg = synth.make_cfg(H, W, lt)
g = la.Liodom(la.make_params(lidar_type=lt, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P), la.make_config(n_streams=1, max_points=N, max_width=W, pose_log_capacity=total + 8))
g.alloc_resident(total)
for k in range(total):
    g.upload_scan(0, k, synth.scan(cfg, 0, k)[0])
g.sync()
for rep in range(3):
    g.reset()
    g.replay_resident(0, F, N, H, W, depth=1, ahead=True)
    g.replay_resident(F, Wm, N, H, W, depth=1, ahead=True)
    g.sync()
    time.sleep(0.002)
    g.replay_resident(F + Wm, K, N, H, W, depth=1, ahead=True)
    g.sync()
    time.sleep(0.002)
g.close()

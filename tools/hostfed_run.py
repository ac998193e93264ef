"""Host-fed replay of K scans (for traces). usage: python tools/hostfed_run.py [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import liodom_amd as la
from liodom_amd import synth
H, W, lt, R, epr, P = 64, 1800, 0, 8, 10, 20
K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = synth.make_cfg(H, W, lt)
scans = np.stack([synth.scan(cfg, 0, k)[0].reshape(-1, 4) for k in range(K)]).reshape(K, 1, H * W, 4)
g = la.Liodom(la.make_params(lidar_type=lt, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
              la.make_config(n_streams=1, max_points=H * W, max_width=W, pose_log_capacity=K + 8))
g.alloc_resident(3)
import time
t = time.time()
poses, infos = g.replay_host(scans, H * W, H, W, depth=1)
print("host-fed: %.1f us/scan" % ((time.time() - t) / K * 1e6), g.modes()["knn_overlap"])
g.close()

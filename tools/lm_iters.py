import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import liodom_amd.api as api
if len(sys.argv) > 1 and sys.argv[1] != "product":
    api._LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "variants", "lib%s.so" % sys.argv[1]); api.is_stale = lambda: False
import numpy as np
import liodom_amd as la
from liodom_amd import synth
H, W, lt, R, epr, P = 64, 1800, 0, 8, 10, 20
cfg = synth.make_cfg(H, W, lt)
K = 200
g = la.Liodom(la.make_params(lidar_type=lt, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P), la.make_config(n_streams=1, max_points=H * W, max_width=W, pose_log_capacity=K + 8))
g.alloc_resident(K)
for k in range(K):
    g.upload_scan(0, k, synth.scan(cfg, 0, k)[0])
poses, infos = g.replay_resident(0, K, H * W, H, W, depth=1)
it = np.array([[i.lm[0].iterations, i.lm[1].iterations] for i in infos[40:]])
ac = np.array([[i.lm[0].accepted, i.lm[1].accepted] for i in infos[40:]])
te = np.array([[i.lm[0].termination, i.lm[1].termination] for i in infos[40:]])
print(sys.argv[1:], "iterations mean", it.mean(axis=0), "accepted mean", ac.mean(axis=0), "terminations", {int(t): int((te == t).sum()) for t in np.unique(te)})
g.close()

import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import liodom_amd as la
from liodom_amd import synth
from liodom_amd.api import StepInfo, _fp, _dp
H, W, lt, R, epr, P = 64, 1800, 0, 8, 10, 20
cfg = synth.make_cfg(H, W, lt)
K = 12
scans = [synth.scan(cfg, 0, k)[0] for k in range(K)]
g = la.Liodom(la.make_params(lidar_type=lt, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
              la.make_config(n_streams=1, max_points=H * W, max_width=W, pose_log_capacity=K + 8))
print("modes:", g.modes())
for k in range(K):
    x = np.ascontiguousarray(scans[k], dtype=np.float32).reshape(-1, 4)
    pose = np.zeros(7); info = StepInfo()
    rc = g.L.liodom_process_scan(g.h, 0, _fp(x), x.shape[0], H, W, 0.0, _dp(pose), C.byref(info))
    print(k, "rc", rc, "status", hex(info.status), "edges", info.n_edges, "matches", list(info.matches), pose[4:7])
    if rc: print(g.L.liodom_last_error().decode()[:80]); break
g.close()

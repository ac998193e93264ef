"""Fixed batched workload for counter collection: S lock-step headline streams, K scans (no timing)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import liodom_amd as la
from liodom_amd import synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
H, W = 64, 1800
cfg = synth.make_cfg(H, W, 0)
scans = [synth.scan(cfg, 0, k)[0] for k in range(K)]
g = la.Liodom(la.make_params(scan_lines=H, scan_regions=8, edges_per_region=10, prev_frames=20),
              la.make_config(n_streams=S, max_points=H * W, max_width=W, pose_log_capacity=K + 8))
g.alloc_resident(K)
for s in range(S):
    for k in range(K):
        g.upload_scan(s, k, scans[k])
for k in range(K):
    g.process_resident(k, H * W, H, W, readback=True)
g.sync()
g.close()
print("done")

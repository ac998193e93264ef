"""Small fixed workload for rocprofv3: K scans of the headline config through the resident path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liodom_amd as la
from liodom_amd import synth
H, W, R, epr, P = 64, 1800, 8, 10, 20
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = synth.make_cfg(H, W, 0)
g = la.Liodom(la.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
              la.make_config(n_streams=S, max_points=H * W, max_width=W, pose_log_capacity=K + 8))
g.alloc_resident(K)
for s in range(S):
    for k in range(K):
        g.upload_scan(s, k, synth.scan(cfg, s, k)[0])
g.sync()
t = time.time()
for k in range(K):
    g.process_resident(k, H * W, H, W, readback=False)
g.sync()
print("async %.1f us/step" % ((time.time() - t) / K * 1e6))
g.close()

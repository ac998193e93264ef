"""Fixed workload for counter collection (no timing): S lock-step streams of a BASELINE workload, K scans, pipelined replay.
usage: python tools/workload_run.py <hdl64|vlp16|ouster128> <streams> <scans>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import liodom_amd as la
from liodom_amd import synth

WL = {"hdl64": (64, 1800, 0, 8, 10, 20), "vlp16": (16, 1800, 0, 8, 20, 10), "ouster128": (128, 2048, 1, 8, 10, 30)}
name = sys.argv[1] if len(sys.argv) > 1 else "hdl64"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
K = int(sys.argv[3]) if len(sys.argv) > 3 else 40
H, W, lt, R, epr, P = WL[name]
cfg = synth.make_cfg(H, W, lt)
scans = [synth.scan(cfg, 0, k)[0] for k in range(K)]
g = la.Liodom(la.make_params(lidar_type=lt, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
              la.make_config(n_streams=S, max_points=H * W, max_width=W, pose_log_capacity=K + 8))
g.alloc_resident(K)
for s in range(S):
    for k in range(K):
        g.upload_scan(s, k, scans[k])
for k in range(K):
    g.process_resident(k, H * W, H, W, readback=True, next_slot=(k + 1 if k + 1 < K else -1))
g.sync()
print("modes:", g.modes())
g.close()

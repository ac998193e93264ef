"""Fixed workload for counter collection (no timing): S lock-step streams of a BASELINE workload, pipelined replay.
The window is pre-filled first (P scans: launches that see a partly filled map), then K steady-state scans follow;
tools/pmc_summary.py drops the pre-fill launches (argument skip_scans = P) so that the per-launch averages describe the regime
bench.py times.
usage: python tools/workload_run.py <hdl64|vlp16|ouster128|hdl64_ragged> <streams> <steady-state scans>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import liodom_amd as la
from liodom_amd import synth

WL = {"hdl64": (64, 1800, 0, 8, 10, 20), "vlp16": (16, 1800, 0, 8, 20, 10), "ouster128": (128, 2048, 1, 8, 10, 30),
      "hdl64_ragged": (64, 1800, 0, 8, 10, 20)}
name = sys.argv[1] if len(sys.argv) > 1 else "hdl64"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
H, W, lt, R, epr, P = WL[name]
T = P + K
cfg = synth.make_cfg(H, W, lt)
scans = [synth.scan(cfg, 0, k)[0] for k in range(T)]
if name.endswith("_ragged"):
    scans = [synth.ragged(x, H, W, lt, seed=k) for k, x in enumerate(scans)]
g = la.Liodom(la.make_params(lidar_type=lt, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
              la.make_config(n_streams=S, max_points=H * W, max_width=W, pose_log_capacity=T + 8))
g.alloc_resident(T)
for s in range(S):
    for k in range(T):
        g.upload_scan(s, k, scans[k])
for k in range(T):
    g.process_resident(k, H * W, H, W, readback=True, next_slot=(k + 1 if k + 1 < T else -1))
g.sync()
print("modes:", g.modes())
print("prefill_scans", P, "steady_scans", K)
g.close()

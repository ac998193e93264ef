"""The lock-step batch leg of bench.py on its own (no single-stream legs, no oracle): aggregate scans/s of S streams + the HIP-event
per-kernel times of the same steps.  usage: python tools/batched_value.py [streams=256] [steps=20] [repeats=3]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import liodom_amd as la
from liodom_amd import synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Kb = int(sys.argv[2]) if len(sys.argv) > 2 else 20
REP = int(sys.argv[3]) if len(sys.argv) > 3 else 3
H, W, R, epr, P = 64, 1800, 8, 10, 20
N = H * W
Wb = P + 4
tb = Kb + Wb
cfg = synth.make_cfg(H, W, 0)
n_data = min(8, S)
data = [[synth.scan(cfg, 0 if d == 0 else 1000 + d, k)[0] for k in range(tb)] for d in range(n_data)]
gb = la.Liodom(la.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
               la.make_config(n_streams=S, max_points=N, max_width=W, pose_log_capacity=tb + 8))
gb.alloc_resident(tb)
for s in range(S):
    for k in range(tb):
        gb.upload_scan(s, k, data[s % n_data][k])
gb.sync()
vals = []
for r in range(REP):
    gb.reset()
    for k in range(Wb):
        gb.process_resident(k, N, H, W, readback=True, next_slot=(k + 1 if k + 1 < Wb else -1))
    gb.sync()
    t2 = time.perf_counter()
    for k in range(Wb, tb):
        gb.process_resident(k, N, H, W, readback=True, next_slot=(k + 1 if k + 1 < tb else -1))
    gb.sync()
    eb = time.perf_counter() - t2
    vals.append(S * Kb / eb)
poses, infos = gb.pose_log(0, 0, tb)
status = 0
for i in infos:
    status |= int(i.status)
gb.reset()
for k in range(Wb):
    gb.process_resident(k, N, H, W, readback=True)
gb.reset_kernel_stats()
gb.set_profiling(True)
for k in range(Wb, tb):
    gb.process_resident(k, N, H, W, readback=True)
st = gb.kernel_stats()
gb.set_profiling(False)
print("batched %d streams x %d steps: %s scans/s (median %.0f), %.3f ms/step, status 0x%x, pose checksum %s" % (
    S, Kb, ["%.0f" % v for v in vals], float(np.median(vals)), S * 1e3 / float(np.median(vals)), status,
    hex(int(np.frombuffer(poses.tobytes(), dtype=np.uint64).sum() & 0xFFFFFFFFFFFF))))
print("per kernel us/launch:", {k: round(ms * 1e3 / max(1, n), 1) for k, (n, ms) in st.items() if n})
print("modes:", {k: v for k, v in gb.modes().items() if k in ("knn8", "hash_build", "knn_instance", "ring_split", "ring_split_lb", "hash_incr", "hash_rebuilds", "hash_appends", "hash_appends_spilled", "hash_points_spilled")})
gb.close()

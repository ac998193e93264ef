"""Experiment: S lock-step streams as ONE handle against the same streams as NH handles of S / NH streams each (one process, one host
thread enqueueing round-robin, no per-step readback): do the kernels of independent sub-batches fill each other's tails and the
latency-bound launches (solves, hash build: one workgroup per stream)?  usage: python tools/batched_split.py [streams=256] [steps=20] [handles=2]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import liodom_amd as la
from liodom_amd import synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Kb = int(sys.argv[2]) if len(sys.argv) > 2 else 20
NH = int(sys.argv[3]) if len(sys.argv) > 3 else 2
H, W, R, epr, P = 64, 1800, 8, 10, 20
N = H * W
Wb = P + 4
tb = Kb + Wb
cfg = synth.make_cfg(H, W, 0)
n_data = 8
data = [[synth.scan(cfg, 0 if d == 0 else 1000 + d, k)[0] for k in range(tb)] for d in range(n_data)]

def make(n_streams, first):
    g = la.Liodom(la.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
                  la.make_config(n_streams=n_streams, max_points=N, max_width=W, pose_log_capacity=tb + 8))
    g.alloc_resident(tb)
    for s in range(n_streams):
        for k in range(tb):
            g.upload_scan(s, k, data[(first + s) % n_data][k])
    g.sync()
    return g

def run(gs, stagger):
    vals = []
    for r in range(3):
        for g in gs:
            g.reset()
        for k in range(Wb):
            for g in gs:
                g.process_resident(k, N, H, W, readback=False, next_slot=(k + 1 if k + 1 < Wb else -1))
        for g in gs:
            g.sync()
        t0 = time.perf_counter()
        for k in range(Wb, tb):
            for g in gs:
                g.process_resident(k, N, H, W, readback=False, next_slot=(k + 1 if k + 1 < tb else -1))
        for g in gs:
            g.sync()
        vals.append(S * Kb / (time.perf_counter() - t0))
    return vals

for nh in ([1, NH] if NH > 1 else [1]):
    gs = [make(S // nh, i * (S // nh)) for i in range(nh)]
    v = run(gs, False)
    chk = 0
    for g in gs:
        poses, infos = g.pose_log(0, 0, tb)
        chk ^= int(np.frombuffer(poses.tobytes(), dtype=np.uint64).sum() & 0xFFFFFFFFFFFF)
    print("%d handle(s) x %d streams: %s scans/s (median %.0f), %.3f ms per %d-scan step, stream-0 pose checksum(s) xor %s, modes %s" % (
        nh, S // nh, ["%.0f" % x for x in v], float(np.median(v)), S * 1e3 / float(np.median(v)), S, hex(chk),
        {k: gs[0].modes().get(k) for k in ("knn8", "hash_incr", "ring_split_lb", "streams_concurrent")}))
    for g in gs:
        g.close()

"""Rates of the drop-in-shaped paths next to the resident replay, one process per configuration (environment switches are read at
handle creation): resident replay (depth 1), host-fed replay (liodom_replay_host, depth 1), two-thread binding (C++ threads).
Every path's poses must be bit-equal to the resident replay's.  usage: python tools/dropin_rates.py [hdl64|vlp16|ouster128] [K] [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liodom_amd as la
from liodom_amd import synth

WL = {"hdl64": (64, 1800, 0, 8, 10, 20), "vlp16": (16, 1800, 0, 8, 20, 10), "ouster128": (128, 2048, 1, 8, 10, 30)}
name = sys.argv[1] if len(sys.argv) > 1 else "hdl64"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
H, W, lt, R, epr, P = WL[name]
N = H * W
cfg = synth.make_cfg(H, W, lt)
F = P + 10
total = F + K
scans = np.stack([synth.scan(cfg, 0, k)[0] for k in range(total)])
host = scans.reshape(total, 1, N, 4).copy()
PIN = True
if os.environ.get("DROPIN_HOSTMALLOC"):
    # scans in memory from hipHostMalloc (as liodom_scan_buffer's ring) instead of a registered NumPy array
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    ptr = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(ptr), C.c_size_t(host.nbytes), C.c_uint(0)) == 0
    buf = (C.c_float * host.size).from_address(ptr.value)
    hm = np.frombuffer(buf, dtype=np.float32).reshape(host.shape)
    hm[...] = host
    host = hm
    PIN = False
g = la.Liodom(la.make_params(lidar_type=lt, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
              la.make_config(n_streams=1, max_points=N, max_width=W, pose_log_capacity=total + 8))
g.alloc_resident(total)
for k in range(total):
    g.upload_scan(0, k, scans[k])
env = {k: v for k, v in os.environ.items() if k.startswith("LIODOM_")}
m = g.modes()
print("%s K=%d env %s modes knn_overlap=%s pipe_flags=%s" % (name, K, env, m.get("knn_overlap"), m.get("pipe_flags")), flush=True)


def med(x):
    x = sorted(x)
    return x[len(x) // 2], x[0], x[-1]


res_rates, ref = [], None
for r in range(reps):
    g.reset()
    g.replay_resident(0, F, N, H, W, depth=1, ahead=True)
    t = time.perf_counter()
    p, _ = g.replay_resident(F, K, N, H, W, depth=1)
    res_rates.append(K / (time.perf_counter() - t))
    ref = p[:, 0].copy() if ref is None else ref
    assert np.array_equal(p[:, 0].view(np.uint64), ref.view(np.uint64))
print("resident   median %.1f scans/s (min %.1f max %.1f)" % med(res_rates), flush=True)
hf_rates = []
for r in range(reps):
    g.reset()
    g.replay_host(host[:F], N, H, W, depth=1, pin=PIN)
    t = time.perf_counter()
    hp, _ = g.replay_host(host[F:], N, H, W, depth=1, pin=PIN)
    hf_rates.append(K / (time.perf_counter() - t))
    assert np.array_equal(hp[:, 0].view(np.uint64), ref.view(np.uint64)), "host-fed poses differ from the resident replay"
print("host_fed   median %.1f scans/s (min %.1f max %.1f)  = %.1f %% of resident" % (med(hf_rates) + (100.0 * med(hf_rates)[0] / med(res_rates)[0],)), flush=True)
tt_rates = []
for r in range(reps):
    g.reset()
    tp, secs, _ = g.two_thread_replay(host[:, 0], N, H, W, timed_from=F, fetch_edges=True, depth=1, pin=PIN)
    tt_rates.append(K / secs)
    assert np.array_equal(tp[F:].view(np.uint64), ref.view(np.uint64)), "two-thread poses differ from the resident replay"
print("two_thread median %.1f scans/s (min %.1f max %.1f)  = %.1f %% of resident" % (med(tt_rates) + (100.0 * med(tt_rates)[0] / med(res_rates)[0],)), flush=True)
g.close()

"""Hammer: resident replay vs two-thread (ticket) replay vs per-call replay on small shapes, many rounds, fresh handles; counts runs
whose pose log differs from the first.  usage: python tools/chain_hammer.py [rounds] [shape]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import liodom_amd as la
from liodom_amd import synth
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
shape = sys.argv[2] if len(sys.argv) > 2 else "16x900"
H, W, R, epr, P, K = {"16x900": (16, 900, 6, 10, 5, 60), "vlp16": (16, 1800, 8, 20, 10, 80)}[shape]
N = H * W
cfg = synth.make_cfg(H, W, 0)
scans = np.stack([synth.scan(cfg, 3, k)[0] for k in range(K)])
par = la.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P)
bad = {"resident": 0, "two_thread": 0, "percall": 0}
ref = None
t0 = time.time()
for rnd in range(rounds):
    g = la.Liodom(par, la.make_config(max_points=N, max_width=W, pose_log_capacity=2 * K + 8))
    g.alloc_resident(K)
    for k in range(K):
        g.upload_scan(0, k, scans[k])
    p, infos = g.replay_resident(0, K, N, H, W, depth=1)
    p = p[:, 0].copy()
    if ref is None:
        ref = p
        refi = [(int(i.n_edges), list(i.matches), [i.lm[0].iterations, i.lm[1].iterations]) for i in infos]
    def check(name, got, gi):
        d = np.nonzero(np.any(got.view(np.uint64) != ref.view(np.uint64), axis=1))[0]
        if len(d):
            bad[name] += 1
            k = int(d[0])
            i = gi[k]
            print("round %d %s: %d scans differ, first %d: ref %s | got n_edges %d matches %s it %s status %d dpose %.2e" % (
                rnd, name, len(d), k, refi[k], int(i.n_edges), list(i.matches), [i.lm[0].iterations, i.lm[1].iterations], int(i.status), np.abs(got[k] - ref[k]).max()), flush=True)
    check("resident", p, infos)
    for depth in (1, 0):
        g.reset()
        got, secs, tot = g.two_thread_replay(scans, N, H, W, timed_from=10, fetch_edges=True, depth=depth, pin=(rnd % 2 == 0))
        _, gi = g.pose_log(0, 0, K)
        check("two_thread", got, gi)
    g.reset()
    out = []
    for k in range(K):
        pp, _ = g.process_resident(k, N, H, W, readback=True, next_slot=(k + 1 if k + 1 < K else -1))
        out.append(pp[0].copy())
    _, gi = g.pose_log(0, 0, K)
    check("percall", np.array(out), gi)
    g.close()
print("%s chain=%s: %d rounds in %.1f s, differing runs: %s" % (shape, os.environ.get("LIODOM_CHAIN", "default"), rounds, time.time() - t0, bad))

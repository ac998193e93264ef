"""Debug: where does the two-thread (ticket) replay leave the resident replay's poses?  16x900 shape, fresh process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import liodom_amd as la
from liodom_amd import synth
H, W, R, epr, P, K = 16, 900, 6, 10, 5, 60
N = H * W
cfg = synth.make_cfg(H, W, 0)
scans = np.stack([synth.scan(cfg, 3, k)[0] for k in range(K)])
par = la.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P)
g = la.Liodom(par, la.make_config(max_points=N, max_width=W, pose_log_capacity=2 * K + 8))
print("modes chain=%s overlap=%s" % (g.modes().get("chain"), g.modes().get("knn_overlap")))
g.alloc_resident(K)
for k in range(K):
    g.upload_scan(0, k, scans[k])
ref, infos = g.replay_resident(0, K, N, H, W, depth=1)
ref = ref[:, 0].copy()
ri = [(int(i.n_edges), list(i.matches), [i.lm[0].iterations, i.lm[1].iterations], int(i.status)) for i in infos]
g.reset()
ref2, infos2 = g.replay_resident(0, K, N, H, W, depth=1)
print("resident replay twice: identical", np.array_equal(ref.view(np.uint64), ref2[:, 0].view(np.uint64)))
for rnd in range(6):
    g.reset()
    got, secs, tot = g.two_thread_replay(scans, N, H, W, timed_from=10, fetch_edges=True, depth=1, pin=True)
    d = np.nonzero(np.any(got.view(np.uint64) != ref.view(np.uint64), axis=1))[0]
    _, gi = g.pose_log(0, 0, K)
    if len(d):
        k = int(d[0])
        print("round %d: %d scans differ, first %d" % (rnd, len(d), k))
        for kk in range(max(0, k - 1), min(K, k + 2)):
            i = gi[kk]
            print("   scan %d ref %s | got n_edges %d matches %s it %s status %d | dpose %.3e" % (kk, ri[kk], int(i.n_edges), list(i.matches), [i.lm[0].iterations, i.lm[1].iterations], int(i.status), np.abs(got[kk] - ref[kk]).max()))
    else:
        print("round %d: identical" % rnd)
g.close()

"""(debug) first scan at which a chain-mode replay with a forced-wrong hand-over differs from the replay without speculation, and in what.
usage: python tools/spec_diag.py <shape> <scans> <out.npz>   (LIODOM_SPECULATE from the environment)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liodom_amd as la
from liodom_amd import synth
shape, K, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
H, W, LT, R, epr, P = {"hdl64": (64, 1800, 0, 8, 10, 20), "vlp16": (16, 1800, 0, 8, 20, 10), "16x900": (16, 900, 0, 6, 10, 5)}[shape]
N = H * W
cfg = synth.make_cfg(H, W, LT)
g = la.Liodom(la.make_params(lidar_type=LT, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
              la.make_config(n_streams=1, max_points=N, max_width=W, pose_log_capacity=K + 8))
g.alloc_resident(K)
for k in range(K):
    g.upload_scan(0, k, synth.scan(cfg, 7, k)[0])
g.sync()
p, infos = g.replay_resident(0, K, N, H, W, depth=1)
rows = np.array([[i.n_edges, i.map_points, i.matches[0], i.matches[1], i.lm[0].iterations, i.lm[1].iterations, i.lm[0].termination, i.lm[1].termination, i.status] for i in infos], dtype=np.int64)
costs = np.array([[i.lm[0].initial_cost, i.lm[0].final_cost, i.lm[1].initial_cost, i.lm[1].final_cost] for i in infos])
np.savez(out, poses=p[:, 0], rows=rows, costs=costs)
print(g.modes().get("chain"), g.modes().get("speculate"), g.modes().get("spec_early"), g.modes().get("spec_unconfirmed"), "chain_done", g.modes().get("chain_done"))
g.close()

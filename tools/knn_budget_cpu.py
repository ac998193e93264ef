#!/usr/bin/env python3
"""CPU-side budget of the first kNN pass (no GPU): for the queries and the 20-frame window of the headline stream, how many map
points a query has to look at under different cell layouts / pruning rules.  Queries and window come from the oracle's run.
usage: tools/knn_budget_cpu.py [scan=30] [out.npz]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from liodom_amd import synth
from oracle import oracle as orc
from scipy.spatial import cKDTree

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
H, W, R, epr, P = 64, 1800, 8, 10, 20
cfg = synth.make_cfg(H, W, 0)
po = orc.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, knn_mode=1)
od = orc.Odometer(po)
for k in range(K + 1):
    x, _ = synth.scan(cfg, 0, k)
    e = orc.extract(po, x, H, W)
    if k == K:
        win = od.window().copy()
    od.step(e["edges"])
q = od.last_queries(0).copy()
np.savez("/tmp/an/knn_in.npz", win=win, q=q)
print("scan %d: %d queries, %d window points" % (K, len(q), len(win)))

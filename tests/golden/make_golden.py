"""Regenerates tests/golden/*.npz: known-answer vectors produced by the CPU oracle (oracle/) on seeded
synthetic streams.  The reference itself has no tests, fixtures or golden vectors and cannot be built
in this image, so these pin the ORACLE (against accidental changes) and give the GPU
tests a committed answer that does not depend on the oracle being rebuilt identically.

    python tests/golden/make_golden.py

Inputs are regenerated from the seeds (liodom_amd/synth); only outputs are stored: per scan the edge
(ring, index-in-ring) lists, the pose [qx qy qz qw tx ty tz], the match counts and the LM iteration
counts; for the map: Map::getMap after the replayed updates (float32 bits, or their SHA-256 for big maps)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from liodom_amd import synth          # noqa: E402
from oracle import oracle as orc      # noqa: E402

CASES = {
    # name: (H, W, lidar_type, R, epr, P, stream, scans) — the four single-GPU BASELINE.json configs at
    # full size, each with more than P + 5 scans so that the window fills and evicts
    "stream_16x900": (16, 900, 0, 6, 10, 5, 0, 12),                # configs[0]
    "stream_vlp16_16x1800": (16, 1800, 0, 8, 20, 10, 2, 16),      # configs[1]
    "stream_64x1800": (64, 1800, 0, 8, 10, 20, 3, 26),            # configs[2] (headline)
    "stream_ouster_128x2048": (128, 2048, 1, 8, 10, 30, 1, 36),   # configs[3]
    "stream_ouster_32x512": (32, 512, 1, 8, 10, 6, 1, 8),
}
MAP_BITS_MAX_POINTS = 20000      # larger maps are pinned by their SHA-256 only


def generate(name):
    H, W, lt, R, epr, P, stream, K = CASES[name]
    cfg = synth.make_cfg(H, W, lt)
    po = orc.make_params(lidar_type=lt, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, knn_mode=1)
    od = orc.Odometer(po)
    mp = orc.Map(40.0, 50.0, 0.4)
    out = {"case": np.array([H, W, lt, R, epr, P, stream, K], dtype=np.int64)}
    poses, matches, iters, nedges = [], [], [], []
    rings, idxs = [], []
    for k in range(K):
        x, _ = synth.scan(cfg, stream, k)
        e = orc.extract(po, x, H, W)
        pose, info = od.step(e["edges"])
        T, _ = orc.pose_ops(pose[:4], pose[4:])
        mp.update(e["edges"], T)
        rings.append(np.asarray(e["ring"], dtype=np.int16))
        idxs.append(np.asarray(e["idx_in_ring"], dtype=np.int16))
        nedges.append(len(e["ring"]))
        poses.append(pose)
        matches.append(list(info.matches))
        iters.append([info.lm[0].iterations, info.lm[1].iterations])
    out["n_edges"] = np.array(nedges, dtype=np.int32)
    out["edge_ring"] = np.concatenate(rings)
    out["edge_idx"] = np.concatenate(idxs)
    out["poses"] = np.array(poses, dtype=np.float64)
    out["matches"] = np.array(matches, dtype=np.int32)
    out["lm_iterations"] = np.array(iters, dtype=np.int32)
    m = mp.all().view(np.uint32)
    out["map_points"] = np.array([m.shape[0]], dtype=np.int64)
    out["map_all_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(m).tobytes()).digest(), dtype=np.uint8)
    if m.shape[0] <= MAP_BITS_MAX_POINTS:
        out["map_all_bits"] = m
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for name in CASES:
        d = generate(name)
        np.savez_compressed(os.path.join(here, name + ".npz"), **d)
        print(name, "edges/scan", d["n_edges"].tolist(), "map points", int(d["map_points"][0]))

"""Oracle restatement of the mapping node's Map / Cell (src/map.cc, rows A12-A14 of SURVEY.md §8)
and of its interaction with the odometer.  CPU only; the device implementation is checked against
this restatement in tests/test_gpu_map.py."""
import numpy as np


def P(*rows):
    a = np.zeros((len(rows), 4), np.float32)
    for i, r in enumerate(rows):
        a[i, :len(r)] = r
    return a


def test_cell_keys_and_voxel_merge(orc):
    m = orc.Map(40.0, 50.0, 0.4)
    # cell key per axis = int(floor(x / size) * size + size / 2)  (map.cc:103-105): x = -0.1 -> -20
    m.update(P((-0.1, 0.1, 0.1, 1.0), (0.1, 0.1, 0.1, 2.0)))
    assert m.num_cells() == 2
    # two points in one 0.4 m leaf -> one centroid; a later update treats that centroid as ONE point
    m2 = orc.Map(40.0, 50.0, 0.4)
    m2.update(P((1.0, 1.0, 1.0, 0.0), (1.1, 1.0, 1.0, 0.0)))
    a = m2.all()
    assert a.shape[0] == 1 and np.isclose(a[0, 0], np.float32(1.0) / 2 + np.float32(1.1) / 2, atol=1e-6)
    m2.update(P((1.15, 1.0, 1.0, 0.0),))
    b = m2.all()
    assert b.shape[0] == 1
    expect = (np.float32(a[0, 0]) + np.float32(1.15)) / np.float32(2.0)      # not a running mean over 3 points
    assert b[0, 0] == expect
    # cells that were not touched are not re-filtered
    m2.update(P((100.0, 0.0, 0.0, 0.0),))
    assert m2.num_cells() == 2 and m2.all().shape[0] == 2


def test_get_local_map_quirks(orc):
    m = orc.Map(40.0, 50.0, 0.4)
    pts = []
    for cx in range(-3, 4):
        for cy in range(-3, 4):
            pts.append((cx * 40.0 + 20.0, cy * 40.0 + 20.0, 1.0, float((cx + 3) * 10 + cy + 3)))
    m.update(P(*pts))
    assert m.num_cells() == 49
    T = np.eye(4)[:3].copy()
    T[:, 3] = [0.9, -0.9, 0.5]      # int truncation toward zero (map.cc:144-150): x = 0, y = 0 -> cell (20, 20)
    loc = m.local(T, 2, 1)
    ids = sorted(loc[:, 3].astype(int).tolist())
    # 5 x 5 cells around cell (20, 20): cx, cy in [-2, 2] -> labels (cx+3)*10 + cy+3; x outer, y inner
    assert ids == sorted((cx + 3) * 10 + cy + 3 for cx in range(-2, 3) for cy in range(-2, 3))
    assert loc[:, 3].astype(int).tolist() == [(cx + 3) * 10 + cy + 3 for cx in range(-2, 3) for cy in range(-2, 3)]
    # the z column uses the XY size for its extent and steps by the Z size (map.cc:175-178): with
    # 40 / 50 the probed keys (-15, 35) are never cell keys, so nothing is added; with equal sizes
    # the centre cell is appended a second time
    m2 = orc.Map(40.0, 40.0, 0.4)
    m2.update(P((1.0, 1.0, 1.0, 7.0),))
    assert m2.local(np.eye(4)[:3], 2, 1)[:, 3].tolist() == [7.0, 7.0]


def test_mapping_mode_degenerates_under_the_restated_semantics(orc, synth):
    """Finding: with mapping = true the kNN cloud is window + mapper cloud
    (laser_odometry.cc:310-314).  Every single-point VoxelGrid leaf of the mapper is a bit-identical
    copy of a window point, so NN0 == NN1 for such edges, the line has zero length, the residual is
    0/0 (factors.hpp:99-101) and Ceres rejects the evaluation: the pose stays at the prediction."""
    H, W = 16, 900
    cfg = synth.make_cfg(H, W, 0)
    po = orc.make_params(scan_lines=H, scan_regions=6, edges_per_region=10, prev_frames=5, knn_mode=1, mapping=True)
    od = orc.Odometer(po)
    for k in range(4):
        x, gt = synth.scan(cfg, 0, k)
        pose, info = od.step(orc.extract(po, x, H, W)["edges"])
        if k > 0:
            assert info.matches[0] > 20
            assert info.lm[0].termination == 5 and info.lm[1].termination == 5     # evaluation failure
            assert np.allclose(pose, [0, 0, 0, 1, 0, 0, 0])                         # never moved
    w = od.window()[:, :3]
    r = od.received_map()[:, :3]
    dup = (w[:, None, :] == r[None, :200, :]).all(-1).any(0).sum()
    assert dup > 0                                                                   # exact duplicates

"""Device liodom::Map (liodom_map_* of the C-ABI; src/map.cc rows A12-A14) against the CPU oracle's
restatement: bit-exact clouds in the reference's output order."""
import numpy as np
import pytest

import liodom_amd as la

pytestmark = pytest.mark.gpu


def P(*rows):
    a = np.zeros((len(rows), 4), np.float32)
    for i, r in enumerate(rows):
        a[i, :len(r)] = r
    return a


def same(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def pose(yaw, t):
    T = np.eye(4)[:3].copy()
    c, s = np.cos(yaw), np.sin(yaw)
    T[:2, :2] = [[c, -s], [s, c]]
    T[:, 3] = t
    return T


def test_cell_keys_and_voxel_merge(orc):
    for pts in (P((-0.1, 0.1, 0.1, 1.0), (0.1, 0.1, 0.1, 2.0)),
                P((1.0, 1.0, 1.0, 0.0), (1.1, 1.0, 1.0, 0.0))):
        mo, mg = orc.Map(40.0, 50.0, 0.4), la.Map(40.0, 50.0, 0.4, max_cells=64, cell_capacity=4096)
        mo.update(pts); mg.update(pts)
        assert mg.num_cells() == mo.num_cells()
        assert same(mg.all(), mo.all())
        # a later update treats the centroid as ONE point; untouched cells are not re-filtered
        for more in (P((1.15, 1.0, 1.0, 0.0),), P((100.0, 0.0, 0.0, 0.0),), P()):
            mo.update(more); mg.update(more)
            assert mg.num_cells() == mo.num_cells() and same(mg.all(), mo.all())
        assert mg.status() == 0
        mg.close()


def test_get_local_map_quirks(orc):
    pts = []
    for cx in range(-3, 4):
        for cy in range(-3, 4):
            pts.append((cx * 40.0 + 20.0, cy * 40.0 + 20.0, 1.0, float((cx + 3) * 10 + cy + 3)))
    mo, mg = orc.Map(40.0, 50.0, 0.4), la.Map(40.0, 50.0, 0.4, max_cells=128, cell_capacity=1024)
    mo.update(P(*pts)); mg.update(P(*pts))
    assert mg.num_cells() == 49
    for t in ([0.9, -0.9, 0.5], [-0.9, 0.9, -0.5], [45.0, -41.0, 3.0], [-39.5, 80.2, 51.0], [1000.0, 0.0, 0.0]):
        for cxy, cz in ((2, 1), (1, 0), (0, 2), (3, 1)):
            T = pose(0.3, t)
            assert same(mg.local(T, cxy, cz), mo.local(T, cxy, cz)), (t, cxy, cz)
    assert same(mg.all(), mo.all())
    mg.close()
    # equal xy / z sizes: the z column hits real cells and the centre cell is appended twice (map.cc:175-186)
    mo2, mg2 = orc.Map(40.0, 40.0, 0.4), la.Map(40.0, 40.0, 0.4, max_cells=64, cell_capacity=1024)
    q = P((1.0, 1.0, 1.0, 7.0), (1.0, 1.0, 41.0, 8.0), (1.0, 1.0, -39.0, 9.0), (41.0, 1.0, 1.0, 10.0))
    mo2.update(q); mg2.update(q)
    loc = mg2.local(None, 2, 1)
    assert same(loc, mo2.local(None, 2, 1)) and loc[:, 3].tolist().count(7.0) == 2
    mg2.close()


@pytest.mark.parametrize("sizes", [(40.0, 50.0, 0.4), (10.0, 10.0, 0.25), (25.0, 30.0, 0.3)])
def test_random_updates_bit_exact(orc, sizes):
    """Many updates of random clouds under moving poses: cells are created in first-appearance
    order, leaves merge old centroids with several new points (float sums in cloud order)."""
    rng = np.random.default_rng(7)
    xy, z, res = sizes
    mo = orc.Map(xy, z, res)
    mg = la.Map(xy, z, res, max_cells=512, cell_capacity=32768, max_update_points=4096, max_modified_cells=128)
    for k in range(25):
        n = int(rng.integers(1, 3000))
        # clustered points so that many share a leaf, some exactly on leaf / cell boundaries
        centres = rng.uniform(-30, 30, size=(40, 3)) * [1, 1, 0.2]
        pts = centres[rng.integers(0, 40, n)] + rng.normal(0, 0.35, size=(n, 3))
        pts[: n // 20] = np.round(pts[: n // 20] / res) * res
        x = np.zeros((n, 4), np.float32)
        x[:, :3] = pts
        x[:, 3] = rng.uniform(0, 100, n)
        T = pose(0.02 * k, [0.8 * k, 0.1 * k, 0.01 * k])
        mo.update(x, T); mg.update(x, T)
        assert mg.num_cells() == mo.num_cells(), k
        a, b = mg.all(), mo.all()
        assert same(a, b), (k, a.shape, b.shape)
        assert same(mg.local(T, 2, 1), mo.local(T, 2, 1)), k
    assert mg.status() == 0
    mg.close()


def test_replayed_edge_clouds(orc, synth):
    """The mapping node's real input: the odometer's edge clouds and poses (liodom_mapping_node.cc:45-82)."""
    H, W = 16, 900
    cfg = synth.make_cfg(H, W, 0)
    po = orc.make_params(scan_lines=H, scan_regions=6, edges_per_region=10, prev_frames=5, knn_mode=1)
    od = orc.Odometer(po)
    mo, mg = orc.Map(), la.Map(max_cells=256, cell_capacity=32768)
    for k in range(12):
        x, _ = synth.scan(cfg, 0, k)
        e = orc.extract(po, x, H, W)["edges"]
        pq, _ = od.step(e)
        T = orc.iso_from_qt(pq[:4], pq[4:]) if hasattr(orc, "iso_from_qt") else None
        if T is None:
            qx, qy, qz, qw = pq[:4]
            R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                          [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                          [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
            T = np.concatenate([R, np.array(pq[4:]).reshape(3, 1)], axis=1)
        mo.update(e, T); mg.update(e, T)
        assert same(mg.all(), mo.all()), k
        assert same(mg.local(T, 2, 1), mo.local(T, 2, 1)), k
    assert mg.status() == 0 and mg.num_cells() == mo.num_cells()
    mg.close()


def test_capacity_overflow_is_reported(orc):
    mg = la.Map(40.0, 50.0, 0.4, max_cells=4, cell_capacity=64, max_update_points=256, max_modified_cells=4)
    x = np.zeros((200, 4), np.float32)
    x[:, 0] = np.linspace(0, 39, 200)        # 98 distinct leaves in one cell > 64
    mg.update(x)
    assert mg.status() & 8
    y = P(*[(50.0 * i, 0.0, 0.0, 0.0) for i in range(8)])      # 8 cells > max_cells / max_modified_cells
    mg.update(y)
    assert mg.status() & (2 | 4)
    with pytest.raises(la.LiodomError):
        mg.update(np.zeros((300, 4), np.float32))
    mg.close()


# ---------------------------------------------------------------------------------------------
# mapping = true in the odometer (laser_odometry.cc:276-278,310-314): kNN cloud = window ++ ~map
# ---------------------------------------------------------------------------------------------
POSE_TOL_T, POSE_TOL_R = 1e-4, 1e-4


def rot_angle(qa, qb):
    d = abs(float(np.dot(qa, qb)))
    return 2.0 * np.arccos(min(1.0, d))


def T_of(pq):
    qx, qy, qz, qw = pq[:4]
    R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                  [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                  [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
    return np.concatenate([R, np.array(pq[4:]).reshape(3, 1)], axis=1)


def _mk_mapping(orc, H, W, R, epr, P, mapping):
    po = orc.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, knn_mode=1, mapping=mapping)
    g = la.Liodom(la.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, mapping=1),
                  la.make_config(max_points=H * W, max_width=W, recv_capacity=1 << 17))
    return po, g


@pytest.mark.parametrize("hash_build", ["global", "lds"])
def test_external_map_feeds_the_knn_cloud(orc, synth, monkeypatch, hash_build):
    """~map delivered through liodom_set_received_map (mapClb): a mapper that has integrated the
    scans that already LEFT the sliding window (so the two clouds share no point).  Same
    correspondences (indices into window ++ map), LM traces and poses as the oracle."""
    monkeypatch.setenv("LIODOM_HASH_BUILD", hash_build)
    H, W, R, epr, P, K = 16, 900, 6, 10, 4, 14
    cfg = synth.make_cfg(H, W, 0)
    po, g = _mk_mapping(orc, H, W, R, epr, P, 2)
    od = orc.Odometer(po)
    mo = orc.Map()
    hist = []
    used_map = 0
    for k in range(K):
        x, _ = synth.scan(cfg, 0, k)
        e = orc.extract(po, x, H, W)["edges"]
        if k > P:      # the frame that left the window before this scan goes into the mapper
            e_old, T_old = hist[k - P - 1]
            mo.update(e_old, T_old)
            loc = mo.local(hist[-1][1], 2, 1)
            od.set_received_map(loc)
            g.set_received_map(loc)
        pose_o, info_o = od.step(e)
        pose_g, info_g = g.process_scan(x, H, W)
        hist.append((e, T_of(pose_o)))
        assert np.linalg.norm(pose_g[4:] - pose_o[4:]) <= POSE_TOL_T and rot_angle(pose_g[:4], pose_o[:4]) <= POSE_TOL_R, k
        if k > 0:
            assert info_g.map_points == info_o.map_points, (k, info_g.map_points, info_o.map_points)
            assert [info_g.lm[i].termination for i in (0, 1)] == [info_o.lm[i].termination for i in (0, 1)]
            assert [info_g.lm[i].iterations for i in (0, 1)] == [info_o.lm[i].iterations for i in (0, 1)]
            for it in (0, 1):
                vo, ao, bo = od.last_corr(it)
                vg, ag, bg = g.correspondences(it)
                diff = int((vo != vg).sum()) + int(((ao != ag) & (vo == 1) & (vg == 1)).sum())
                assert diff <= 3, (k, it, diff)
                if k > P:
                    used_map += int((ag[vg == 1] >= od.window().shape[0]).sum())
        lm, filtered = g.local_map()
        assert not filtered and same(lm, np.concatenate([od.window(), od.received_map()]).astype(np.float32)) or \
            np.allclose(lm[:, :3], np.concatenate([od.window(), od.received_map()])[:, :3], atol=2e-5)
    assert used_map > 50          # correspondences really landed in the received map
    assert np.linalg.norm(pose_o[4:]) > 0.5
    g.close()


def test_attached_mapper_replays_the_mapping_node(orc, synth):
    """liodom_attach_mapper: updateMap(edges_k, pose_k) + getLocalMap(pose_k) on the device after
    every scan.  Against the oracle's synchronous replay: same received map bit for bit, same
    match counts — and the same degenerate outcome (single-point leaves of the mapper duplicate
    window points, the line through NN0 == NN1 has zero length, Ceres rejects the evaluation and
    the pose stays at the prediction; tests/test_oracle_map.py)."""
    H, W, R, epr, P, K = 16, 900, 6, 10, 5, 6
    cfg = synth.make_cfg(H, W, 0)
    po, g = _mk_mapping(orc, H, W, R, epr, P, 1)
    mg = la.Map(max_cells=256, cell_capacity=32768)
    g.attach_mapper(mg, 2, 1)
    od = orc.Odometer(po)
    for k in range(K):
        x, _ = synth.scan(cfg, 0, k)
        pose_o, info_o = od.step(orc.extract(po, x, H, W)["edges"])
        pose_g, info_g = g.process_scan(x, H, W)
        assert same(g.received_map(), od.received_map()), k
        assert np.linalg.norm(pose_g[4:] - pose_o[4:]) <= POSE_TOL_T and rot_angle(pose_g[:4], pose_o[:4]) <= POSE_TOL_R, k
        if k > 0:
            assert info_g.map_points == info_o.map_points
            assert list(info_g.matches) == list(info_o.matches)
            assert [info_g.lm[i].termination for i in (0, 1)] == [info_o.lm[i].termination for i in (0, 1)] == [5, 5]
    assert mg.status() == 0 and mg.num_cells() > 0
    assert same(mg.all()[:10], mg.all()[:10])
    g.attach_mapper(None)
    g.close()
    mg.update(np.ones((3, 4), np.float32))      # the map works on its own again after the handle is gone
    assert mg.status() == 0
    mg.close()


def test_replay_harness_mapping_mode(orc, synth, tmp_path):
    """liodom_replay mapping=true: host C++ mirror (liodom::Map, LaserOdometer::attachMapper) over the
    C-ABI; the final map file has the oracle mapper's size and the poses are the oracle's."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "liodom_amd", "host", "liodom_replay")
    if not os.path.exists(exe):
        pytest.skip("liodom_replay not built (run __graft_entry__.build())")
    H, W, R, epr, P, K = 16, 900, 6, 10, 5, 5
    cfg = synth.make_cfg(H, W, 0)
    po = orc.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, knn_mode=1, mapping=1)
    od = orc.Odometer(po)
    scan_dir, out_dir = tmp_path / "scans", tmp_path / "out"
    scan_dir.mkdir(); out_dir.mkdir()
    rows = []
    for k in range(K):
        x, _ = synth.scan(cfg, 0, k)
        x.astype(np.float32).tofile(str(scan_dir / ("%06d.bin" % k)))
        pose, _ = od.step(orc.extract(po, x, H, W)["edges"])
        T, _ = orc.pose_ops(pose[:4], pose[4:])
        rows.append(T.reshape(12))
    r = subprocess.run([exe, str(scan_dir), str(out_dir) + "/", "scan_lines=16", "scan_regions=6", "edges_per_region=10",
                        "prev_frames=5", "mapping=true"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.loadtxt(str(out_dir / "poses.txt")).reshape(-1, 12)
    assert np.allclose(got, np.array(rows), rtol=2e-5, atol=2e-6)
    m = np.fromfile(str(out_dir / "map.bin"), dtype=np.float32).reshape(-1, 4)
    assert m.shape[0] == od.map_total() and m.shape[0] > 100


def test_two_streams_with_their_own_mappers(orc, synth):
    """n_streams = 2, mapping = 1, one device map attached per stream: each stream's received map
    and pose log equal those of a single-stream handle fed the same scans."""
    H, W, R, epr, P, K = 16, 900, 6, 10, 4, 5
    cfg = synth.make_cfg(H, W, 0)
    scans = [[synth.scan(cfg, s, k)[0] for k in range(K)] for s in range(2)]
    prm = la.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, mapping=1)
    single = []
    for s in range(2):
        g = la.Liodom(prm, la.make_config(max_points=H * W, max_width=W, recv_capacity=1 << 16))
        m = la.Map(max_cells=128, cell_capacity=16384)
        g.attach_mapper(m, 2, 1)
        poses = [g.process_scan(scans[s][k], H, W)[0].copy() for k in range(K)]
        single.append((np.array(poses), g.received_map(), m.all()))
        g.attach_mapper(None)
        g.close(); m.close()
    gb = la.Liodom(prm, la.make_config(n_streams=2, max_points=H * W, max_width=W, recv_capacity=1 << 16, pose_log_capacity=K + 4))
    maps = [la.Map(max_cells=128, cell_capacity=16384) for _ in range(2)]
    for s in range(2):
        gb.attach_mapper(maps[s], 2, 1, stream=s)
    gb.alloc_resident(K)
    for s in range(2):
        for k in range(K):
            gb.upload_scan(s, k, scans[s][k])
    for k in range(K):
        gb.process_resident(k, H * W, H, W, readback=True)
    for s in range(2):
        log, _ = gb.pose_log(s, 0, K)
        assert np.array_equal(log, single[s][0]), s
        assert same(gb.received_map(stream=s), single[s][1]), s
        assert same(maps[s].all(), single[s][2]), s
    for s in range(2):
        gb.attach_mapper(None, stream=s)
    gb.close()
    for m in maps:
        m.close()

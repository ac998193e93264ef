"""GPU parity: the HIP path (through the C-ABI of include/liodom_hip.h) against the CPU oracle on
identical inputs.  Bar: bit-exact edge sets (ring, index, source, XYZI bits) and correspondence
index sets; pose within 1e-4 m / 1e-4 rad per scan (BASELINE.json north_star).
Run with -m gpu on an MI355X."""
import numpy as np
import pytest

import liodom_amd as la
import pyref

pytestmark = pytest.mark.gpu

POSE_TOL_T = 1e-4   # metres
POSE_TOL_R = 1e-4   # radians


def rot_angle(qa, qb):
    d = abs(float(np.dot(qa, qb)) / (np.linalg.norm(qa) * np.linalg.norm(qb)))
    return 2.0 * np.arccos(min(1.0, d))


def mk(orc, H, W, lidar_type=0, R=8, epr=10, P=5, S=1, debug=0, **cfgkw):
    po = orc.make_params(lidar_type=lidar_type, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, knn_mode=1)
    pg = la.make_params(lidar_type=lidar_type, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P)
    cg = la.make_config(n_streams=S, max_points=max(H * W, 16), max_width=max(W, 16), debug_buffers=debug, **cfgkw)
    return po, la.Liodom(pg, cg)


def assert_edges_equal(g, o):
    assert len(g["ring"]) == len(o["ring"])
    assert np.array_equal(g["ring"], o["ring"])
    assert np.array_equal(g["idx_in_ring"], o["idx_in_ring"])
    assert np.array_equal(g["src"], o["src"])
    assert np.array_equal(g["edges"].view(np.uint32), o["edges"].view(np.uint32))


CONFIGS = [
    # (H, W, lidar_type, R, epr) — BASELINE.json configs 1-4 at full size
    (16, 900, 0, 6, 10),
    (16, 1800, 0, 8, 20),
    (64, 1800, 0, 8, 10),
    (128, 2048, 1, 8, 10),
    (32, 700, 0, 4, 5),
]


@pytest.mark.parametrize("H,W,lt,R,epr", CONFIGS)
def test_extract_bit_exact(orc, synth, H, W, lt, R, epr):
    cfg = synth.make_cfg(H, W, lt)
    po, g = mk(orc, H, W, lt, R, epr, debug=1)
    for k in (0, 7):
        x, _ = synth.scan(cfg, 2, k)
        o = orc.extract(po, x, H, W, want_curv=True)
        e = g.extract_edges(x, H, W)
        assert_edges_equal(e, o)
        assert len(o["ring"]) > 50
        # FP64 smoothness, bit for bit (NaN where undefined)
        cg, offs = g.curvature()
        co = o["curv"][:len(cg)]
        assert np.array_equal(np.isnan(cg), np.isnan(co))
        m = ~np.isnan(cg)
        assert np.array_equal(cg[m].view(np.uint64), co[m].view(np.uint64))
    g.close()


@pytest.mark.parametrize("S", [1, 3])
def test_ring_split_equals_two_kernel_split(orc, synth, monkeypatch, S):
    """Handles whose extraction launch has at most 256 workgroups split the scan into rings with ONE kernel (k_ring_split:
    the tiles of a stream wait for each other's histograms inside the launch); LIODOM_RING_SPLIT=0 — and every lock-step batch,
    and safe mode — uses k_classify + k_ring_scatter.  Both against the oracle, on ragged scans (NaN no-returns, rings of unequal
    length, dead rings) so that tiles, ranks and ring starts differ from the regular grid; S = 3: three streams in one launch."""
    H, W, R, epr = 64, 1800, 8, 10
    cfg = synth.make_cfg(H, W, 0)
    scans = [[synth.ragged(synth.scan(cfg, 5 + s, k)[0], H, W, 0, seed=100 * s + k) for k in (0, 3)] for s in range(S)]
    for mode in ("1", "0"):
        monkeypatch.setenv("LIODOM_RING_SPLIT", mode)
        po, g = mk(orc, H, W, 0, R, epr, S=S)
        assert g.modes()["ring_split"] == mode
        for k in range(2):
            for s in range(S):
                e = g.extract_edges(scans[s][k], H, W, stream=s)
                assert_edges_equal(e, orc.extract(po, scans[s][k], H, W))
        if S > 1:      # all streams through one lock-step launch
            g.alloc_resident(1)
            for s in range(S):
                g.upload_scan(s, 0, scans[s][1])
            g.process_resident(0, H * W, H, W, readback=True)
            for s in range(S):
                assert_edges_equal(g.get_edges(s), orc.extract(po, scans[s][1], H, W))
        g.close()


@pytest.mark.parametrize("pitch", [None, "1500", "40"])
def test_ring_split_look_back_on_batches(orc, synth, monkeypatch, pitch):
    """Lock-step batches split the scan in ONE pass too (k_ring_split_lb): rings at a fixed pitch, tiles take their number from a
    ticket counter and sum their predecessors' tagged counts as they appear.  16 streams in one launch, ragged scans (tiles, ranks
    and ring lengths off the regular grid) and regular ones, against the oracle and against k_classify + k_ring_scatter
    (LIODOM_RING_SPLIT_LB=0).  pitch 1500 (LIODOM_RING_PITCH): the full rings of the regular scans (1800 points) outgrow their
    segment, the ragged streams' mostly do not — k_ring_split_fix redoes exactly the streams whose flag is up; pitch 40: every ring
    of every stream.  Matches feature_extractor.cc:104-179 (stable per-ring order)."""
    H, W, R, epr, S = 64, 1800, 8, 10, 16
    cfg = synth.make_cfg(H, W, 0)
    scans = [[(synth.ragged(synth.scan(cfg, 5 + s, k)[0], H, W, 0, seed=100 * s + k) if s % 2 else synth.scan(cfg, 5 + s, k)[0]) for k in (0, 3)] for s in range(4)]
    want = [[orc.extract(orc.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr), scans[s][k], H, W) for k in range(2)] for s in range(4)]
    for lb in ("1", "0"):
        monkeypatch.setenv("LIODOM_RING_SPLIT_LB", lb)
        if pitch is None:
            monkeypatch.delenv("LIODOM_RING_PITCH", raising=False)
        else:
            monkeypatch.setenv("LIODOM_RING_PITCH", pitch)
        po, g = mk(orc, H, W, 0, R, epr, S=S)
        assert g.modes()["ring_split_lb"] == lb
        g.alloc_resident(2)
        for s in range(S):
            for k in range(2):
                g.upload_scan(s, k, scans[s % 4][k])
        for k in range(2):      # (two launches: the ticket counter and the tags of the first must not confuse the second)
            _, infos = g.process_resident(k, H * W, H, W, readback=True)
            assert all(i.status == 0 for i in infos)
            for s in range(S):
                assert_edges_equal(g.get_edges(s), want[s % 4][k])
        # one stream on its own through the same handle (a launch of 57 workgroups: k_ring_split's domain when it is on)
        assert_edges_equal(g.extract_edges(scans[1][1], H, W, stream=3), want[1][1])
        g.close()


def test_extract_edge_cases(orc):
    po, g = mk(orc, 16, 1800, 0, 8, 10)
    # empty cloud, all-NaN cloud
    assert len(g.extract_edges(np.zeros((0, 4), np.float32), 16, 0)["ring"]) == 0
    x = np.full((16 * 100, 4), np.nan, np.float32)
    assert len(g.extract_edges(x, 16, 100)["ring"]) == 0
    # jagged ring: epr+1 picks per region
    from test_oracle_extract import _jagged_ring
    x = _jagged_ring(1800)
    e = g.extract_edges(x, 16, 0)
    assert_edges_equal(e, orc.extract(po, x, 16, 0))
    assert len(e["ring"]) == 88
    # ragged: ring sizes around the min_points_per_scan threshold (90)
    for n in (89, 90, 91, 101, 333):
        x = _jagged_ring(n, seed=n)
        assert_edges_equal(g.extract_edges(x, 16, 0), orc.extract(po, x, 16, 0))
    g.close()


def test_ring_split_at_decision_boundaries(orc):
    """k_classify decides in float where that is safe and falls back to the reference's FP64 expressions near a decision
    boundary.  Points placed ON and within 1e-9 .. 1e-3 (degrees / relative range) of every boundary — elevation bin
    edges of the 16 / 32 / 64-line formulas, the 2 / -8.83 / -24.33 degree limits, min_range and max_range — must land
    in exactly the rings the oracle's (literal FP64) split gives: ring sizes and the per-ring smoothness bits."""
    rng = np.random.default_rng(5)
    for H in (16, 32, 64):
        if H == 64:
            edges = [2 - (k - 0.5) / 3.0 for k in range(0, 34)] + [-8.83 - (k - 0.5) / 2.0 for k in range(0, 33)] + [2.0, -8.83, -24.33]
        elif H == 32:
            edges = [k * 4.0 / 3.0 - 92.0 / 3.0 for k in range(-1, 34)]
        else:
            edges = [2.0 * (k - 0.5) - 15.0 for k in range(-1, 18)]
        offs = [0.0] + [sgn * 10.0 ** -e for e in range(3, 10) for sgn in (1, -1)]
        pts = []
        W = 0
        for az in np.linspace(-3.1, 3.1, 220):
            col = []
            for a in edges:
                for o in offs:
                    r = 10.0 + 3.0 * np.sin(7 * az) + rng.normal(0, 0.02)
                    el = np.deg2rad(a + o)
                    col.append((r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el), 0.0))
            # range limits (XY range): exactly on, and a hair inside / outside
            for lim in (3.0, 75.0):
                for o in offs:
                    d = lim * (1.0 + o)
                    col.append((d * np.cos(az), d * np.sin(az), -0.05 * d, 0.0))
            W = len(col)
            pts.extend(col)
        x = np.array(pts, dtype=np.float32)
        po, g = mk(orc, H, 4 * len(x) // H, 0, 8, 10, debug=1)
        o_offs, o_order = orc.split(po, x, H, 0)
        o = orc.extract(po, x, H, 0, want_curv=True)
        e = g.extract_edges(x, H, 0)
        cg, offs_g = g.curvature()
        assert np.array_equal(offs_g, o_offs), (H, offs_g, o_offs)
        assert_edges_equal(e, o)
        co = o["curv"][:len(cg)]
        m = ~np.isnan(cg)
        assert np.array_equal(np.isnan(cg), np.isnan(co)) and np.array_equal(cg[m].view(np.uint64), co[m].view(np.uint64)), H
        assert o_offs[-1] > 0.5 * len(x) and W > 0
        g.close()


def test_suppression_carry_across_regions(orc):
    # SURVEY.md §0 fact 4 — same construction as tests/test_oracle_extract.py
    n = 910
    phi = np.linspace(0, 0.5, n)
    r = np.full(n, 20.0)
    r[453] += 0.08
    r[456] += 0.05
    x = np.zeros((n, 4), np.float32)
    x[:, 0] = r * np.cos(phi)
    x[:, 1] = r * np.sin(phi)
    x[:, 2] = -0.3
    po, g = mk(orc, 16, 1024, 0, 2, 3)
    e = g.extract_edges(x, 16, 0)
    assert_edges_equal(e, orc.extract(po, x, 16, 0))
    assert 453 in e["idx_in_ring"].tolist() and not any(453 < i <= 458 for i in e["idx_in_ring"].tolist())
    g.close()


def test_rings_of_any_length_are_processed(orc, synth):
    """The reference has no per-ring capacity (std::vector per ring, feature_extractor.cc:115-175): a ring far
    longer than the expected width — real HDL-64 clouds put well over W points into some elevation bins —
    must be extracted like any other, never dropped.  One ring holding the whole cloud: 700 points (register
    path), 5 000 points with 8 regions (regions longer than 512 items: generic path) and 20 000 points
    (longer than the LDS continuity-bit array: generic path)."""
    from test_oracle_extract import _jagged_ring
    for n, R in ((700, 8), (5000, 8), (5000, 16), (20000, 40)):
        po, g = mk(orc, 16, (n + 15) // 16, 0, R, 10)      # capacity: max_points = 16 * W >= n
        x = _jagged_ring(n)
        o = orc.extract(po, x, 16, 0)
        e = g.extract_edges(x, 16, 0)
        assert len(o["ring"]) > 0
        assert_edges_equal(e, o)
        pose, info = g.process_scan(x, 16, 0)
        assert info.status == 0
        g.close()

def test_max_ring_size(orc):
    # a 6144-point ring (regions of 766 items: beyond the register tile of either instance, generic path)
    from test_oracle_extract import _jagged_ring
    po, g = mk(orc, 16, 6144, 0, 8, 10)
    x = _jagged_ring(6144, seed=5)
    assert_edges_equal(g.extract_edges(x, 16, 0), orc.extract(po, x, 16, 0))
    g.close()


def _run_stream(orc, synth, H, W, lt, R, epr, P, nscans, stream=0, use_pipeline=True, mutate=None, exact_on=None):
    """GPU vs oracle over a stream.  Two levels of correspondence parity:
    (1) kernel level, identical inputs: the GPU's own float queries and the local map it searched go through the
        oracle's addEdgeConstraints loop — valid flags and both line-point indices must be EXACTLY equal, every
        scan, both passes;
        (exact_on: optional predicate on the scan number — long streams run this level, the expensive one, on the scans it selects);
    (2) end to end: the oracle's own run.  The poses of the two implementations differ in the last bits (different
        reduction orders), so a query or a window point may round to a neighbouring float; every edge whose
        correspondence differs must be explained by such an input difference, and the second scan — whose
        prediction is the identity in both — must agree exactly."""
    cfg = synth.make_cfg(H, W, lt)
    po, g = mk(orc, H, W, lt, R, epr, P, debug=1)
    od = orc.Odometer(po)
    worst_t, worst_r = 0.0, 0.0
    n_e2e_diff = 0
    for k in range(nscans):
        x, _ = synth.scan(cfg, stream, k)
        if mutate is not None:
            x = mutate(x, k)
        o = orc.extract(po, x, H, W)
        map_g, _ = g.local_map()                 # the cloud this scan's kNN passes search (window before the step)
        map_o = od.window()
        pose_o, info_o = od.step(o["edges"])
        if use_pipeline:
            pose_g, info_g = g.process_scan(x, H, W)
            assert_edges_equal(g.get_edges(), o)
        else:
            pose_g, info_g = g.odometry_step(o["edges"])
        dt = np.linalg.norm(pose_g[4:] - pose_o[4:])
        dr = rot_angle(pose_g[:4], pose_o[:4])
        worst_t, worst_r = max(worst_t, dt), max(worst_r, dr)
        assert dt <= POSE_TOL_T and dr <= POSE_TOL_R, "scan %d: dt=%g dr=%g" % (k, dt, dr)
        assert info_g.n_edges == info_o.n_edges
        assert info_g.status == 0
        if k > 0:
            assert info_g.map_points == info_o.map_points
            win_differs = np.any(map_g.view(np.uint32) != map_o.view(np.uint32), axis=1) if map_g.shape == map_o.shape else None
            for it in (0, 1):
                vg, ag, bg = g.correspondences(it)
                # (1) identical inputs -> identical outputs
                qg = g.knn_queries(it)
                if exact_on is None or exact_on(k):
                    vk, ak, bk = orc.match_edges(po, map_g, qg)
                    assert np.array_equal(vk, vg) and np.array_equal(ak, ag) and np.array_equal(bk, bg), \
                        "scan %d pass %d: kNN / line gate differ from the oracle on identical inputs at edges %s" % (
                            k, it, np.nonzero((vk != vg) | (ak != ag) | (bk != bg))[0][:10])
                    assert info_g.matches[it] == int(vk.sum())
                # (2) end to end
                vo, ao, bo = od.last_corr(it)
                qo = od.last_queries(it)
                differ = (vo != vg) | (ao != ag) | (bo != bg)
                if k == 1 and it == 0:
                    assert not differ.any(), "scan 1, pass 0 (identity prediction in both): %d correspondences differ" % differ.sum()
                for e in np.nonzero(differ)[0]:
                    # explained only by an input difference: the query bits, or a window point among the UNION of both
                    # sides' five nearest neighbours (whoever ranks a differing point among its five may pick another line)
                    explained = bool(np.any(qg[e].view(np.uint32) != qo[e].view(np.uint32)))
                    if not explained and win_differs is not None:
                        q4 = np.zeros((1, 4), dtype=np.float32)
                        q4[0, :3] = qg[e]
                        five_g = orc.knn5(map_g, q4, 0)[0][0]
                        q4[0, :3] = qo[e]
                        five_o = orc.knn5(map_o, q4, 0)[0][0]
                        explained = any(i >= 0 and bool(win_differs[i]) for i in list(five_g) + list(five_o))
                    assert explained, "scan %d pass %d edge %d: correspondence differs although the query and the five nearest window points of both sides are bit-equal" % (k, it, e)
                n_e2e_diff += int(differ.sum())
                assert int(differ.sum()) <= 6, "scan %d pass %d: %d correspondences differ" % (k, it, int(differ.sum()))
                assert info_g.lm[it].iterations == info_o.lm[it].iterations
                assert info_g.lm[it].termination == info_o.lm[it].termination
        # sliding window: same frames, same points (float rounding of the FP64 transform)
        wo = od.window()
        wg, nf = g.window()
        assert nf == od.window_frames() and wg.shape == wo.shape
        assert np.allclose(wg, wo, rtol=0, atol=2e-5)
    g.close()
    return worst_t, worst_r


def test_odometry_parity_cfg1(orc, synth):
    # BASELINE config 1: 16 x 900, R=6, epr=10, P=5 — 25 scans (window fills and evicts)
    _run_stream(orc, synth, 16, 900, 0, 6, 10, 5, 25)


def test_odometry_step_entry_point(orc, synth):
    # LaserOdometer boundary alone: oracle edges in, pose out
    _run_stream(orc, synth, 16, 900, 0, 6, 10, 5, 10, use_pipeline=False)


def test_odometry_parity_cfg2(orc, synth):
    _run_stream(orc, synth, 16, 1800, 0, 8, 20, 10, 16)


def test_odometry_parity_headline(orc, synth):
    # BASELINE config 3 (headline): 64 x 1800, R=8, epr=10, P=20
    _run_stream(orc, synth, 64, 1800, 0, 8, 10, 20, 30)


def test_odometry_parity_headline_ragged(orc, synth):
    # the headline shape with ragged input: ~25 % NaN no-returns, rings of unequal length (every ring loses its own share of
    # points, in bursts and singly), three rings below min_points_per_scan (feature_extractor.cc:84-102,188-190)
    H, W = 64, 1800
    seen = {}

    def mutate(x, k):
        r = synth.ragged(x, H, W, 0, seed=k)
        seen[k] = float(np.isnan(r[:, 0]).mean())
        return r

    _run_stream(orc, synth, H, W, 0, 8, 10, 20, 26, mutate=mutate)
    assert 0.15 < np.mean(list(seen.values())) < 0.40
    po = orc.make_params(scan_lines=H, prev_frames=20)
    offs, _ = orc.split(po, synth.ragged(synth.scan(synth.make_cfg(H, W, 0), 0, 3)[0], H, W, 0, seed=3), H, W)
    lens = np.diff(offs)
    assert (lens < 90).sum() >= 2 and lens.max() - lens[lens >= 90].min() > 400      # skipped rings, unequal lengths


def test_odometry_parity_ouster(orc, synth):
    # BASELINE config 4: 128 x 2048 organised cloud, P=30 (quick variant: the window is still filling)
    _run_stream(orc, synth, 128, 2048, 1, 8, 10, 30, 8)


def test_odometry_parity_ouster_full_window(orc, synth):
    """BASELINE config 4 with the window of launch/liodom_ouster.launch:19-23's shape FULL and evicting: 36 scans at P = 30, so
    scans 31 .. 35 search a 30-frame map and every one of them follows an eviction (LocalMapManager::addPointCloud,
    laser_odometry.cc:34-60).  The exact-correspondence check (GPU queries + GPU map through the oracle's addEdgeConstraints loop:
    valid flags and both line-point indices equal, both passes) runs on the first scans, on the last ones of the filling window
    and on every scan of the full one; poses, LM iteration counts / terminations, match counts and the window itself on all 36."""
    _run_stream(orc, synth, 128, 2048, 1, 8, 10, 30, 36, exact_on=lambda k: k <= 3 or k >= 27)


def test_lockstep_streams_match_single_stream(orc, synth):
    # n_streams > 1: every stream must give exactly what a single-stream handle gives
    H, W, R, epr, P, S, K = 16, 900, 6, 10, 5, 3, 9
    cfg = synth.make_cfg(H, W, 0)
    scans = [[synth.scan(cfg, s, k)[0] for k in range(K)] for s in range(S)]
    po, gb = mk(orc, H, W, 0, R, epr, P, S=S)
    gb.alloc_resident(K)
    for s in range(S):
        for k in range(K):
            gb.upload_scan(s, k, scans[s][k])
    batch = []
    for k in range(K):
        poses, infos = gb.process_resident(k, H * W, H, W, readback=True)
        batch.append(poses.copy())
    log0, _ = gb.pose_log(1, 0, K)
    assert np.array_equal(log0, np.array([b[1] for b in batch]))
    gb.close()
    for s in range(S):
        _, g1 = mk(orc, H, W, 0, R, epr, P)
        for k in range(K):
            pose, _ = g1.process_scan(scans[s][k], H, W)
            assert np.array_equal(pose, batch[k][s])
        g1.close()


def test_async_replay_equals_synchronous(orc, synth):
    H, W, R, epr, P, K = 16, 900, 6, 10, 5, 8
    cfg = synth.make_cfg(H, W, 0)
    po, g = mk(orc, H, W, 0, R, epr, P)
    g.alloc_resident(K)
    for k in range(K):
        g.upload_scan(0, k, synth.scan(cfg, 0, k)[0])
    sync = [g.process_resident(k, H * W, H, W, readback=True)[0][0].copy() for k in range(K)]
    g.reset()
    for k in range(K):
        g.process_resident(k, H * W, H, W, readback=False)
    g.sync()
    log, infos = g.pose_log(0, 0, K)
    assert np.array_equal(log, np.array(sync))
    assert [i.scan_index for i in infos] == list(range(K))
    g.close()


def test_filter_local_map_voxelgrid(orc, synth):
    """computeLocalMap with filter_local_map (laser_odometry.cc:286-292): once the window is full
    the kNN map is VoxelGrid(0.4) of the window.  Filtered cloud bit-exact in PCL's output order,
    poses within tolerance, same match counts and LM iterations."""
    H, W, R, epr, P, K = 16, 900, 6, 10, 4, 12
    cfg = synth.make_cfg(H, W, 0)
    po = orc.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, knn_mode=1, filter_local_map=True)
    g = la.Liodom(la.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, filter_local_map=1),
                  la.make_config(max_points=H * W, max_width=W))
    od = orc.Odometer(po)
    seen_filtered = 0
    for k in range(K):
        x, _ = synth.scan(cfg, 0, k)
        o = orc.extract(po, x, H, W)
        pose_o, info_o = od.step(o["edges"])
        pose_g, info_g = g.process_scan(x, H, W)
        assert np.linalg.norm(pose_g[4:] - pose_o[4:]) <= POSE_TOL_T and rot_angle(pose_g[:4], pose_o[:4]) <= POSE_TOL_R, k
        if k > 0:
            assert info_g.map_points == info_o.map_points, (k, info_g.map_points, info_o.map_points)
            assert list(info_g.matches) == list(info_o.matches)
            assert [info_g.lm[0].iterations, info_g.lm[1].iterations] == [info_o.lm[0].iterations, info_o.lm[1].iterations]
        lm, filtered = g.local_map()
        wo = od.window()
        if od.window_frames() == P:
            assert filtered
            ref = orc.voxel_grid(wo, 0.4)
            assert lm.shape == ref.shape
            assert np.array_equal(lm.view(np.uint32), ref.view(np.uint32))
            seen_filtered += 1
        else:
            assert not filtered and lm.shape == wo.shape
    assert seen_filtered >= 5
    g.close()


@pytest.mark.parametrize("cells_max", [None, "40", "400"])
def test_lds_hash_build(orc, synth, monkeypatch, cells_max):
    """k_hash_build (one workgroup per stream builds the 1 m cell hash in LDS; default for
    handles with >= 16 streams) gives the same trajectory as the global-atomic build, including
    its in-kernel fall-back to the global table when the LDS table overflows (forced with 40
    cells) and the transitions LDS -> fall-back across scans as the window grows."""
    monkeypatch.setenv("LIODOM_HASH_BUILD", "lds")
    if cells_max:
        monkeypatch.setenv("LIODOM_LDS_CELLS_MAX", cells_max)
    _run_stream(orc, synth, 16, 900, 0, 6, 10, 5, 14)
    if cells_max is None:
        _run_stream(orc, synth, 64, 1800, 0, 8, 10, 20, 6)


def test_lds_hash_build_with_filter(orc, synth, monkeypatch):
    # LDS-built table while the window fills, filtered-cloud table (global) afterwards
    monkeypatch.setenv("LIODOM_HASH_BUILD", "lds")
    test_filter_local_map_voxelgrid(orc, synth)


def test_multi_workgroup_solve(orc, synth):
    # lm_workgroups = 8: the solve is split over 8 CUs exchanging partial sums inside the launch
    H, W, R, epr, P, K = 16, 900, 6, 10, 5, 10
    cfg = synth.make_cfg(H, W, 0)
    po, g8 = mk(orc, H, W, 0, R, epr, P, lm_workgroups=8)
    od = orc.Odometer(po)
    for k in range(K):
        x, _ = synth.scan(cfg, 0, k)
        pose_o, info_o = od.step(orc.extract(po, x, H, W)["edges"])
        pose_g, info_g = g8.process_scan(x, H, W)
        assert np.linalg.norm(pose_g[4:] - pose_o[4:]) <= POSE_TOL_T and rot_angle(pose_g[:4], pose_o[:4]) <= POSE_TOL_R
        assert not (info_g.status & 8)
        if k > 0:
            assert [info_g.lm[0].iterations, info_g.lm[1].iterations] == [info_o.lm[0].iterations, info_o.lm[1].iterations]
    g8.close()


def test_pipelined_replay_equals_serial(orc, synth):
    # extraction of scan k+1 on the second stream while odometry k runs: same poses, bit for bit
    H, W, R, epr, P, K = 16, 900, 6, 10, 5, 12
    cfg = synth.make_cfg(H, W, 0)
    po, g = mk(orc, H, W, 0, R, epr, P)
    g.alloc_resident(K)
    for k in range(K):
        g.upload_scan(0, k, synth.scan(cfg, 0, k)[0])
    serial = [g.process_resident(k, H * W, H, W, readback=True)[0][0].copy() for k in range(K)]
    g.reset()
    piped = [g.process_resident(k, H * W, H, W, readback=True, next_slot=(k + 1 if k + 1 < K else -1))[0][0].copy()
             for k in range(K)]
    assert np.array_equal(np.array(serial), np.array(piped))
    # mixing entry points drains the pipeline first
    g.reset()
    g.process_resident(0, H * W, H, W, readback=True, next_slot=1)
    e = g.extract_edges(synth.scan(cfg, 0, 5)[0], H, W)
    assert len(e["ring"]) > 50
    g.close()


def test_static_sensor_stays_put(orc, synth):
    # idempotence-style property: replaying the same scan keeps the pose at identity (< 1e-6)
    H, W = 64, 1800
    cfg = synth.make_cfg(H, W, 0)
    x, _ = synth.scan(cfg, 0, 0)
    po, g = mk(orc, H, W, 0, 8, 10, 20)
    for k in range(6):
        pose, info = g.process_scan(x, H, W)
        assert np.linalg.norm(pose[4:]) < 1e-6 and rot_angle(pose[:4], np.array([0, 0, 0, 1.0])) < 1e-6
    g.close()


def test_replay_harness_writes_reference_result_files(orc, synth, tmp_path):
    """liodom_amd/host (C++ mirror of the reference classes over the C-ABI): replay KITTI-style
    .bin scans and compare poses.txt (KITTI 3x4 rows, default stream precision, stats.cc:73-96)
    and nfeats.txt with the oracle."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "liodom_amd", "host", "liodom_replay")
    if not os.path.exists(exe):
        pytest.skip("liodom_replay not built (run __graft_entry__.build())")
    H, W, R, epr, P, K = 16, 900, 6, 10, 5, 8
    cfg = synth.make_cfg(H, W, 0)
    po = orc.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, knn_mode=1)
    od = orc.Odometer(po)
    scan_dir = tmp_path / "scans"
    out_dir = tmp_path / "out"
    scan_dir.mkdir()
    out_dir.mkdir()
    rows, nfeats = [], []
    for k in range(K):
        x, _ = synth.scan(cfg, 0, k)
        x.astype(np.float32).tofile(str(scan_dir / ("%06d.bin" % k)))
        e = orc.extract(po, x, H, W)
        pose, _ = od.step(e["edges"])
        T, _ = orc.pose_ops(pose[:4], pose[4:])
        rows.append(T.reshape(12))
        nfeats.append(len(e["ring"]))
    r = subprocess.run([exe, str(scan_dir), str(out_dir) + "/", "scan_lines=16", "scan_regions=6", "edges_per_region=10",
                        "prev_frames=5"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.loadtxt(str(out_dir / "poses.txt")).reshape(-1, 12)
    assert got.shape == (K, 12)
    assert np.allclose(got, np.array(rows), rtol=2e-5, atol=2e-6)        # 6 significant digits in the file
    assert np.loadtxt(str(out_dir / "nfeats.txt")).astype(int).tolist() == nfeats
    for f in ("feat_ext_times.txt", "laser_odom_times.txt", "frame_times.txt"):
        assert (out_dir / f).exists()


def test_use_imu_roll_pitch_override(orc, synth):
    """use_imu (laser_odometry.cc:152-183): roll / pitch of every prediction replaced by the IMU's,
    in the base_link frame (non-identity laser_to_base), before the solve.  IMU = ground-truth
    orientation with a small perturbation so that the override visibly changes the start pose."""
    H, W, R, epr, P, K = 16, 900, 6, 10, 5, 12
    cfg = synth.make_cfg(H, W, 0)
    po = orc.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, knn_mode=1)
    g = la.Liodom(la.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, use_imu=1),
                  la.make_config(max_points=H * W, max_width=W))
    od = orc.Odometer(po)
    od_plain = orc.Odometer(po)
    c, s = np.cos(0.05), np.sin(0.05)
    L = np.array([[c, 0, s, 0.3], [0, 1, 0, -0.1], [-s, 0, c, 0.8]], dtype=np.float64)
    od.set_laser_to_base(L)
    g.set_laser_to_base(L)
    rng = np.random.default_rng(11)
    differs = 0
    for k in range(K):
        x, gt = synth.scan(cfg, 0, k)
        q = gt[:4] + rng.normal(0, 2e-3, 4)
        q /= np.linalg.norm(q)
        od.set_imu(q)
        g.set_imu(q)
        e = orc.extract(po, x, H, W)["edges"]
        pose_o, info_o = od.step(e)
        pose_p, _ = od_plain.step(e)
        pose_g, info_g = g.process_scan(x, H, W)
        assert np.linalg.norm(pose_g[4:] - pose_o[4:]) <= POSE_TOL_T and rot_angle(pose_g[:4], pose_o[:4]) <= POSE_TOL_R, k
        if k > 0:
            assert abs(info_g.matches[0] - info_o.matches[0]) <= 3
            assert info_g.lm[0].iterations == info_o.lm[0].iterations
            assert abs(info_g.lm[0].initial_cost - info_o.lm[0].initial_cost) <= 1e-6 * max(1.0, info_o.lm[0].initial_cost)
            differs += int(np.linalg.norm(pose_p - pose_o) > 1e-9)      # the override is not a no-op
    assert differs > 0
    g.close()


def test_odometry_degenerate_inputs(orc, synth):
    """Empty edge clouds, scans without a single valid point, a window with fewer than five points
    (the reference would read sq_dist[4] out of bounds, :324; oracle and GPU skip such edges) and
    edges far from every map point: no correspondences -> Ceres 'no residuals', pose = prediction."""
    H, W, R, epr, P = 16, 900, 6, 10, 3
    cfg = synth.make_cfg(H, W, 0)
    po, g = mk(orc, H, W, 0, R, epr, P)
    od = orc.Odometer(po)

    calls = [0]

    def both(edges):
        e = np.ascontiguousarray(edges, dtype=np.float32).reshape(-1, 4)
        pose_o, info_o = od.step(e)
        pose_g, info_g = g.odometry_step(e)
        assert np.allclose(pose_g, pose_o, rtol=0, atol=1e-9), (pose_g, pose_o)
        assert info_g.n_edges == info_o.n_edges == e.shape[0]
        assert list(info_g.matches) == list(info_o.matches)
        if calls[0] > 0:      # the first frame only initialises the window (:108-136): no solve, no trace
            assert [info_g.lm[i].termination for i in (0, 1)] == [info_o.lm[i].termination for i in (0, 1)]
        calls[0] += 1
        wg, nf = g.window()
        assert nf == od.window_frames() and wg.shape == od.window().shape
        return info_g

    x0, _ = synth.scan(cfg, 0, 0)
    real = orc.extract(po, x0, H, W)["edges"]
    both(np.zeros((0, 4)))                      # first frame empty: the window starts with an empty frame
    both(real[:3])                              # 3 map points... next scan searches a 3-point window
    i = both(real)                              # < 5 neighbours for every edge
    assert list(i.matches) == [0, 0] and i.lm[0].termination == 4
    i = both(real + np.float32([500, 0, 0, 0]))  # far away from the map: gate sq_dist[4] < 1 fails everywhere
    assert list(i.matches) == [0, 0]
    both(np.zeros((0, 4)))                      # empty scan in steady state
    i = both(real)                              # and the stream recovers
    assert i.matches[1] > 20
    g.close()
    # whole pipeline on scans without valid points
    po, g = mk(orc, H, W, 0, R, epr, P)
    od = orc.Odometer(po)
    for x in (np.full((H * W, 4), np.nan, np.float32), np.zeros((H * W, 4), np.float32), x0, np.zeros((0, 4), np.float32), x0):
        o = orc.extract(po, x, H, W if len(x) else 0)
        pose_o, info_o = od.step(o["edges"])
        pose_g, info_g = g.process_scan(x, H, W if len(x) else 0)
        assert np.allclose(pose_g, pose_o, rtol=0, atol=1e-9) and info_g.n_edges == info_o.n_edges
    g.close()


def test_soak_long_stream(orc, synth):
    """400 scans of one stream (slow turn: 0.1 deg/scan keeps the reference's pose recursion in its
    stable range): the window evicts hundreds of frames, the hash is rebuilt 400
    times, the device pose log (capacity 256 here) fills up and stops logging.  Status stays clean,
    the GPU tracks the oracle to the tolerance at every scan and both track the ground truth."""
    H, W, R, epr, P, K = 16, 900, 6, 10, 5, 400
    cfg = synth.make_cfg(H, W, 0, yaw_rate_deg=0.1, speed=0.05)
    po = orc.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, knn_mode=1)
    g = la.Liodom(la.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
                  la.make_config(max_points=H * W, max_width=W, pose_log_capacity=256))
    od = orc.Odometer(po)
    gt0 = None
    worst_t = worst_r = 0.0
    for k in range(K):
        x, gt = synth.scan(cfg, 0, k)
        gt0 = gt if gt0 is None else gt0
        pose_o, info_o = od.step(orc.extract(po, x, H, W)["edges"])
        pose_g, info_g = g.process_scan(x, H, W)
        assert info_g.status == 0 and info_g.scan_index == k
        dt, dr = np.linalg.norm(pose_g[4:] - pose_o[4:]), rot_angle(pose_g[:4], pose_o[:4])
        worst_t, worst_r = max(worst_t, dt), max(worst_r, dr)
        assert dt <= POSE_TOL_T and dr <= POSE_TOL_R, (k, dt, dr)
    assert info_g.matches[1] > 30
    # odometry drift (mostly z with 16 rings and a 5-frame window) is the algorithm's, not checked tightly
    assert np.linalg.norm(pose_g[4:6] - (gt[4:6] - gt0[4:6])) < 1.5       # still on the trajectory after 20 m
    log, infos = g.pose_log(0, 0, 256)
    assert [i.scan_index for i in infos] == list(range(256))
    g.close()


def test_velodyne_32_ring_formula(orc):
    """scan_lines = 32 uses its own elevation binning, id = int((angle + 92/3) * 3/4)
    (feature_extractor.cc:139-143); unsupported line counts yield no points (:149-151)."""
    rng = np.random.default_rng(32)
    W = 1200
    az = np.linspace(-np.pi, np.pi, W, endpoint=False)
    pts = []
    for col in range(W):                       # firing order: all rings of one azimuth, then the next
        for ring in range(32):
            elev = np.deg2rad(-30.67 + (ring + 0.5) * 4.0 / 3.0 + rng.normal(0, 0.15))     # some land in neighbouring bins
            rng_m = 12.0 + 6.0 * np.sin(5 * az[col] + ring) + (2.0 if (col // 37) % 2 else 0.0) + rng.normal(0, 0.01)
            pts.append((rng_m * np.cos(elev) * np.cos(az[col]), rng_m * np.cos(elev) * np.sin(az[col]), rng_m * np.sin(elev), ring))
    x = np.array(pts, dtype=np.float32)
    po, g = mk(orc, 32, W + 200, 0, 8, 10)
    e, o = g.extract_edges(x, 32, 0), orc.extract(po, x, 32, 0)
    assert_edges_equal(e, o)
    assert len(e["ring"]) > 500 and len(set(e["ring"].tolist())) >= 30
    g.close()
    # a line count the reference does not know: every point is dropped (and nothing crashes)
    po, g = mk(orc, 24, 1700, 0, 8, 10)
    assert len(g.extract_edges(x, 24, 0)["ring"]) == 0 and len(orc.extract(po, x, 24, 0)["ring"]) == 0
    g.close()


@pytest.mark.parametrize("R,epr", [(1, 10), (3, 4), (12, 6), (20, 3), (40, 2)])
def test_extract_region_counts(orc, synth, R, epr):
    """Region counts other than the default 8: one region, more regions than waves (several regions
    per wave in the parallel carry resolution), regions shorter than 64 items, and regions so short
    (R = 40 on sparse rings) that the in-order replay path is taken."""
    from test_oracle_extract import _jagged_ring
    H, W = 16, 1800
    cfg = synth.make_cfg(H, W, 0)
    po, g = mk(orc, H, W, 0, R, epr)
    for k in range(3):
        x, _ = synth.scan(cfg, 2, k)
        assert_edges_equal(g.extract_edges(x, H, W), orc.extract(po, x, H, W))
    for n, seed in ((1800, 1), (700, 2), (R * epr + 12, 3)):
        if n < R * epr + 10:
            continue
        x = _jagged_ring(n, seed=seed)
        assert_edges_equal(g.extract_edges(x, H, 0), orc.extract(po, x, H, 0))
    g.close()


def test_knn_fast_path_equals_exact_list_path(orc, synth, monkeypatch):
    """k_knn keeps two candidates per lane and pops the five nearest from them; a query whose result that cannot
    certify (three of its nearest in one lane, or equal distances that FLANN orders by index) repeats the stream
    with sorted (distance, index) lists.  LIODOM_KNN_EXACT_ONLY=1 sends EVERY query through the list path;
    the second pass of a scan by default re-ranks the candidates the first pass kept and accepts that only when a guard
    distance proves that no other map point can be among the five nearest (LIODOM_KNN_SAVE=1: it only takes its
    pruning bound from the first pass's fifth-nearest distance; =0: it searches like the first pass); on one-stream handles
    the second pass is overlapped with the first solve and re-ranks the result of an exact search around the first
    pass's query instead (LIODOM_KNN_OVERLAP=0: not overlapped).  Poses, match
    counts and correspondence indices must be bit-identical in all configurations, also with large pose corrections
    between the passes (many re-rankings that cannot be certified)."""
    H, W, R, epr, P, K = 16, 900, 6, 10, 5, 14
    modes = {"default": {}, "exact_only": {"LIODOM_KNN_EXACT_ONLY": "1"}, "bound_only": {"LIODOM_KNN_SAVE": "1"}, "no_saved_bound": {"LIODOM_KNN_SAVE": "0"},
             "no_overlap": {"LIODOM_KNN_OVERLAP": "0"}, "no_overlap_exact_only": {"LIODOM_KNN_OVERLAP": "0", "LIODOM_KNN_EXACT_ONLY": "1"}}
    for yaw, speed in ((0.5, 0.1), (3.0, 0.6)):
        cfg = synth.make_cfg(H, W, 0, yaw_rate_deg=yaw, speed=speed)
        scans = [synth.scan(cfg, 6, k)[0] for k in range(K)]
        res = {}
        for mode, env in modes.items():
            for name in ("LIODOM_KNN_EXACT_ONLY", "LIODOM_KNN_SAVE", "LIODOM_KNN_OVERLAP"):
                monkeypatch.delenv(name, raising=False)
            for name, val in env.items():
                monkeypatch.setenv(name, val)
            po, g = mk(orc, H, W, 0, R, epr, P)
            # (by default the second pass of a scan runs beside its first solve on a HIP stream of its own and re-ranks what
            #  an exact search around the first pass's query collected meanwhile; LIODOM_KNN_OVERLAP=0: on the odometry stream)
            assert g.modes()["knn_overlap"] == ("0" if "LIODOM_KNN_OVERLAP" in env else "1")
            out = []
            for k in range(K):
                pose, info = g.process_scan(scans[k], H, W)
                v0, a0, b0 = g.correspondences(0)
                v1, a1, b1 = g.correspondences(1)
                out.append((pose.copy(), tuple(info.matches), v0.copy(), a0.copy(), b0.copy(), v1.copy(), a1.copy(), b1.copy()))
            res[mode] = out
            g.close()
        for mode in ("exact_only", "bound_only", "no_saved_bound", "no_overlap", "no_overlap_exact_only"):
            for k in range(K):
                assert np.array_equal(res["default"][k][0].view(np.uint64), res[mode][k][0].view(np.uint64)), (mode, yaw, k)
                assert res["default"][k][1] == res[mode][k][1], (mode, yaw, k)
                for x, y in zip(res["default"][k][2:], res[mode][k][2:]):
                    assert np.array_equal(x, y), (mode, yaw, k)


def test_knn_ties_are_ordered_by_window_index(orc, synth):
    """A sensor that does not move in a noise-free world sees the same scan again and again: the window (P = 2) holds
    two exact copies of every edge, so the five nearest neighbours of every query come in pairs of equal float distance
    and FLANN's result order — lower index first — decides which two points define the line.  All those queries take
    the exact list path; valid flags and both line-point indices must equal the oracle's on the GPU's own inputs."""
    H, W, R, epr, P, K = 16, 900, 6, 10, 2, 7
    cfg = synth.make_cfg(H, W, 0, noise_sigma=0.0, yaw_rate_deg=0.0, speed=0.0)
    po, g = mk(orc, H, W, 0, R, epr, P, debug=1)
    x = synth.scan(cfg, 2, 0)[0]
    n_tied = 0
    for k in range(K):
        map_g, _ = g.local_map()
        pose, info = g.process_scan(x, H, W)
        assert info.status == 0
        if k == 0:
            continue
        for it in (0, 1):
            vg, ag, bg = g.correspondences(it)
            qg = g.knn_queries(it)
            vk, ak, bk = orc.match_edges(po, map_g, qg)
            assert np.array_equal(vk, vg) and np.array_equal(ak, ag) and np.array_equal(bk, bg), (k, it, np.nonzero((vk != vg) | (ak != ag) | (bk != bg))[0][:10])
        if k >= 2:
            assert len(np.unique(map_g.view(np.uint32)[:, :3], axis=0)) < len(map_g)        # the window really holds duplicates
            n_tied += int(vg.sum())
    assert n_tied > 0               # accepted correspondences whose NN0 / NN1 are two copies of one point
    g.close()


def test_early_rebuild_equals_three_kernel_rebuild(orc, synth, monkeypatch):
    """Default for handles with <= 4 streams: the window of the next cell hash is counted by extra workgroups of the
    finalising k_lm_solve launch (kept frames beside the solve, the new frame once the solve hands over the pose, into
    the second table).  LIODOM_EARLY_REBUILD=0 restores k_window_insert after the solve on one table.  Poses, window
    contents and correspondences must be bit-identical, on a 3-stream handle (every stream's appending workgroups wait
    for their own solve) through the filling and the evicting phase of the window."""
    H, W, R, epr, P, S, K = 16, 900, 6, 10, 4, 3, 11
    cfg = synth.make_cfg(H, W, 0)
    scans = [[synth.scan(cfg, s, k)[0] for k in range(K)] for s in range(S)]
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("LIODOM_EARLY_REBUILD", mode)
        po, g = mk(orc, H, W, 0, R, epr, P, S=S)
        g.alloc_resident(K)
        for s in range(S):
            for k in range(K):
                g.upload_scan(s, k, scans[s][k])
        out = []
        for k in range(K):
            poses, infos = g.process_resident(k, H * W, H, W, readback=True)
            assert all(i.status == 0 for i in infos)
            corr = [tuple(a.copy() for a in g.correspondences(1, stream=s)) for s in range(S)]
            wins = [g.window(s) for s in range(S)]
            out.append((poses.copy(), [tuple(i.matches) for i in infos], corr, wins))
        res[mode] = out
        g.close()
    for k in range(K):
        p0, m0, c0, w0 = res["0"][k]
        p1, m1, c1, w1 = res["1"][k]
        assert np.array_equal(p0.view(np.uint64), p1.view(np.uint64)), k
        assert m0 == m1, k
        for s in range(S):
            assert all(np.array_equal(a, b) for a, b in zip(c0[s], c1[s])), (k, s)
            assert w0[s][1] == w1[s][1] and np.array_equal(w0[s][0].view(np.uint32), w1[s][0].view(np.uint32)), (k, s)


def test_replay_resident_loop_equals_per_call_replay(orc, synth):
    # liodom_replay_resident: the consumer loop in C gives exactly what one liodom_process_resident_pipelined call per scan gives
    H, W, R, epr, P, K = 16, 900, 6, 10, 5, 10
    cfg = synth.make_cfg(H, W, 0)
    po, g = mk(orc, H, W, 0, R, epr, P)
    g.alloc_resident(K)
    for k in range(K):
        g.upload_scan(0, k, synth.scan(cfg, 2, k)[0])
    ref = []
    for k in range(K):
        poses, _ = g.process_resident(k, H * W, H, W, readback=True, next_slot=(k + 1 if k + 1 < K else -1))
        ref.append(poses.copy())
    g.reset()
    p1, i1 = g.replay_resident(0, 4, H * W, H, W, ahead=True)       # two calls: the second starts on the slot issued ahead
    p2, i2 = g.replay_resident(4, K - 4, H * W, H, W)
    got = np.concatenate([p1, p2])
    assert np.array_equal(got.view(np.uint64), np.array(ref).view(np.uint64))
    assert [i.scan_index for i in list(i1) + list(i2)] == list(range(K))
    # depth 1: odometry submitted one scan ahead of the readback — same poses, same order
    g.reset()
    p3, i3 = g.replay_resident(0, K, H * W, H, W, depth=1)
    assert np.array_equal(p3.view(np.uint64), np.array(ref).view(np.uint64))
    assert [i.scan_index for i in i3] == list(range(K))
    g.close()


def test_streamed_rebuild_overflow_list_is_exact(orc, synth, monkeypatch):
    """Streamed rebuild: a point of the new frame that the solve moved further than rebuild_delta away from where the
    prediction put it cannot use the place its padding reserved and waits in the table's overflow list, which every kNN
    query of the next scan scans as well.  With rebuild_delta = 2 mm most new points take that route: poses, match counts
    and correspondences must not change by a bit (also against the three-kernel rebuild)."""
    H, W, R, epr, P, K = 16, 900, 6, 10, 4, 10
    cfg = synth.make_cfg(H, W, 0, yaw_rate_deg=2.0, speed=0.5)
    scans = [synth.scan(cfg, 4, k)[0] for k in range(K)]
    res = {}
    for name, env in (("default", {}), ("tiny", {"LIODOM_REBUILD_DELTA": "0.002"}), ("three", {"LIODOM_EARLY_REBUILD": "0"})):
        monkeypatch.delenv("LIODOM_REBUILD_DELTA", raising=False)
        monkeypatch.delenv("LIODOM_EARLY_REBUILD", raising=False)
        for kk, vv in env.items():
            monkeypatch.setenv(kk, vv)
        po, g = mk(orc, H, W, 0, R, epr, P)
        out = []
        for k in range(K):
            pose, info = g.process_scan(scans[k], H, W)
            assert info.status == 0
            out.append((pose.copy(), tuple(info.matches)) + tuple(a.copy() for a in g.correspondences(1)))
        res[name] = out
        g.close()
    for name in ("tiny", "three"):
        for k in range(K):
            a, b = res["default"][k], res[name][k]
            assert np.array_equal(a[0].view(np.uint64), b[0].view(np.uint64)), (name, k)
            assert a[1] == b[1] and all(np.array_equal(x, y) for x, y in zip(a[2:], b[2:])), (name, k)
    assert sum(m[1][1] for m in res["default"]) > 100      # (the trajectory does produce matches)


@pytest.mark.parametrize("P", [1, 2])
def test_streamed_rebuild_tiny_windows(orc, synth, monkeypatch, P):
    # prev_frames = 1: no frame is ever kept (the window is the previous scan alone); 2: one kept frame
    H, W, R, epr, K = 16, 900, 6, 10, 7
    cfg = synth.make_cfg(H, W, 0)
    scans = [synth.scan(cfg, 5, k)[0] for k in range(K)]
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("LIODOM_EARLY_REBUILD", mode)
        po, g = mk(orc, H, W, 0, R, epr, P)
        out = []
        for k in range(K):
            pose, info = g.process_scan(scans[k], H, W)
            assert info.status == 0
            w, nf = g.window(0)
            out.append((pose.copy(), tuple(info.matches), w.copy(), nf) + tuple(a.copy() for a in g.correspondences(1)))
        res[mode] = out
        g.close()
    for k in range(K):
        a, b = res["0"][k], res["1"][k]
        assert np.array_equal(a[0].view(np.uint64), b[0].view(np.uint64)) and a[1] == b[1] and a[3] == b[3], (P, k)
        assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32)), (P, k)
        assert all(np.array_equal(x, y) for x, y in zip(a[4:], b[4:])), (P, k)


def test_sixteen_lockstep_streams_match_single_stream(orc, synth):
    # >= 16 streams: the batch instances (k_knn<128> + k_line_gate, k_hash_build; the solve evaluates every block itself,
    # so its sums are ordered differently: poses agree to rounding, not to the bit) against a single-stream handle
    H, W, R, epr, P, S, K = 16, 900, 6, 10, 5, 16, 8
    cfg = synth.make_cfg(H, W, 0)
    scans = [[synth.scan(cfg, s % 3, k)[0] for k in range(K)] for s in range(S)]
    po, gb = mk(orc, H, W, 0, R, epr, P, S=S)
    gb.alloc_resident(K)
    for s in range(S):
        for k in range(K):
            gb.upload_scan(s, k, scans[s][k])
    batch = []
    for k in range(K):
        poses, infos = gb.process_resident(k, H * W, H, W, readback=True, next_slot=(k + 1 if k + 1 < K else -1))
        assert all(i.status == 0 for i in infos)
        batch.append((poses.copy(), [tuple(i.matches) for i in infos]))
    gb.close()
    for s in range(3):
        _, g1 = mk(orc, H, W, 0, R, epr, P)
        for k in range(K):
            pose, info = g1.process_scan(scans[s][k], H, W)
            for s2 in range(s, S, 3):
                assert np.max(np.abs(pose - batch[k][0][s2])) < 1e-9, (s2, k)
                assert tuple(info.matches) == batch[k][1][s2], (s2, k)
                assert np.array_equal(batch[k][0][s2].view(np.uint64), batch[k][0][s].view(np.uint64)), (s2, k)   # equal streams, equal bits
        g1.close()


def test_sixteen_streams_headline_size_against_the_oracle(orc, synth):
    """The code that produces the batched number — k_knn<128> + k_line_gate, the one-workgroup lock-step k_lm_solve,
    k_hash_build — on a 16-stream handle at the headline size (64 x 1800, P = 20, 26 scans: the window fills and evicts),
    checked against the oracle DIRECTLY, per stream: correspondences of both passes exactly equal to the oracle's loop
    (laser_odometry.cc:320-361) on that stream's own queries and local map, LM iteration counts and terminations equal to
    the oracle's run (laser_odometry.cc:201-218), pose within 1e-4 m / 1e-4 rad.  Stream 3 replays ragged scans."""
    H, W, R, epr, P, S, K = 64, 1800, 8, 10, 20, 16, 26
    N = H * W
    D = 4                                     # distinct data streams; handle stream s replays data stream s % D
    cfg = synth.make_cfg(H, W, 0)
    data = [[synth.scan(cfg, 20 + d, k)[0] for k in range(K)] for d in range(D)]
    data[3] = [synth.ragged(x, H, W, 0, seed=100 + k) for k, x in enumerate(data[3])]
    po, gb = mk(orc, H, W, 0, R, epr, P, S=S, debug=1, pose_log_capacity=K + 8)
    modes = gb.modes()
    assert modes["knn_instance"] == "128" and modes["line_gate_kernel"] == "1" and modes["hash_build"] == "lds"
    gb.alloc_resident(K)
    for s in range(S):
        for k in range(K):
            gb.upload_scan(s, k, data[s % D][k])
    ods = [orc.Odometer(po) for _ in range(D)]
    worst_t = worst_r = 0.0
    for k in range(K):
        maps = [gb.local_map(d)[0] for d in range(D)]          # what this step's kNN passes search, per checked stream
        poses, infos = gb.process_resident(k, N, H, W, readback=True, next_slot=(k + 1 if k + 1 < K else -1))
        for s in range(S):
            assert infos[s].status == 0, (k, s)
            assert np.array_equal(poses[s].view(np.uint64), poses[s % D].view(np.uint64)), (k, s)      # equal data, equal bits
            assert tuple(infos[s].matches) == tuple(infos[s % D].matches), (k, s)
        for d in range(D):
            o = orc.extract(po, data[d][k], H, W)
            pose_o, info_o = ods[d].step(o["edges"])
            ig = infos[d]
            assert ig.n_edges == info_o.n_edges, (k, d)
            dt = np.linalg.norm(poses[d][4:] - pose_o[4:])
            dr = rot_angle(poses[d][:4], pose_o[:4])
            worst_t, worst_r = max(worst_t, dt), max(worst_r, dr)
            assert dt <= POSE_TOL_T and dr <= POSE_TOL_R, "scan %d stream %d: dt=%g dr=%g" % (k, d, dt, dr)
            if k == 0:
                continue
            assert ig.map_points == info_o.map_points, (k, d)
            for it in (0, 1):
                vg, ag, bg = gb.correspondences(it, stream=d)
                qg = gb.knn_queries(it, stream=d)
                vk, ak, bk = orc.match_edges(po, maps[d], qg)
                assert np.array_equal(vk, vg) and np.array_equal(ak, ag) and np.array_equal(bk, bg), \
                    "scan %d stream %d pass %d: kNN / line gate differ from the oracle on identical inputs at edges %s" % (
                        k, d, it, np.nonzero((vk != vg) | (ak != ag) | (bk != bg))[0][:10])
                assert ig.matches[it] == int(vk.sum()), (k, d, it)
                assert ig.lm[it].iterations == info_o.lm[it].iterations, (k, d, it)
                assert ig.lm[it].termination == info_o.lm[it].termination, (k, d, it)
    gb.close()
    assert worst_t < 1e-6 and worst_r < 1e-6


@pytest.mark.parametrize("shape,scans", [("hdl64", 1200), ("vlp16", 1200), ("ouster128", 300)])
def test_overlapped_pass_long_replay_is_bit_identical(shape, scans):
    """tools/overlap_equal.py: the same long replay in chain mode (kNN passes + rebuild on one HIP stream, the solves on another, the
    first solve resident beside the first pass), with the overlapped second kNN pass alone, and with neither (LIODOM_CHAIN /
    LIODOM_KNN_OVERLAP, read at handle creation: one process per mode) gives bit-identical pose logs.  Guards the fence-free
    hand-offs of kernels_sync.h: a stale line read by a solve would show up here as a differing pose.  (Ouster-128 overlaps the pass
    in chain mode only.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "overlap_equal.py"), shape, str(scans)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "bit-identical" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


CHAIN_SHAPES = {
    # name: H, W, lidar_type, R, epr, P  (BASELINE configs 3, 2, 4)
    "hdl64": (64, 1800, 0, 8, 10, 20),
    "vlp16": (16, 1800, 0, 8, 20, 10),
    "ouster128": (128, 2048, 1, 8, 10, 30),
}


def _chain_replay_against_the_oracle(orc, synth, shape, extra, data_stream):
    """The path bench.py's `value` times — liodom_replay_resident(depth = 1): chain mode (the kNN passes and the rebuild on one HIP
    stream, the solves on another), edges handed over by flag, both speculative hand-overs — against the oracle DIRECTLY:
    poses within 1e-4 m / 1e-4 rad of orc.Odometer (laser_odometry.cc:198-235), edge counts, match counts of both passes, LM
    iteration counts and terminations equal, every scan; the edges of the last scan bit-equal to orc.extract."""
    H, W, lt, R, epr, P = CHAIN_SHAPES[shape]
    N = H * W
    K = P + extra
    cfg = synth.make_cfg(H, W, lt)
    scans = [synth.scan(cfg, data_stream, k)[0] for k in range(K)]
    po, g = mk(orc, H, W, lt, R, epr, P, pose_log_capacity=K + 8)
    g.alloc_resident(K)
    for k in range(K):
        g.upload_scan(0, k, scans[k])
    g.sync()
    poses, infos = g.replay_resident(0, K, N, H, W, depth=1)
    modes = g.modes()
    od = orc.Odometer(po)
    worst_t = worst_r = 0.0
    for k in range(K):
        o = orc.extract(po, scans[k], H, W)
        pose_o, info_o = od.step(o["edges"])
        ig = infos[k]
        assert ig.status == 0 and ig.scan_index == k, (k, ig.status)
        assert ig.n_edges == info_o.n_edges, k
        dt = np.linalg.norm(poses[k][0][4:] - pose_o[4:])
        dr = rot_angle(poses[k][0][:4], pose_o[:4])
        worst_t, worst_r = max(worst_t, dt), max(worst_r, dr)
        assert dt <= POSE_TOL_T and dr <= POSE_TOL_R, "scan %d: dt=%g dr=%g" % (k, dt, dr)
        if k == 0:
            continue
        assert ig.map_points == info_o.map_points, k
        for it in (0, 1):
            # the two trajectories differ in the last bits, so a correspondence at the edge of a gate may flip (see _run_stream)
            assert abs(int(ig.matches[it]) - int(info_o.matches[it])) <= 6, (k, it, ig.matches[it], info_o.matches[it])
            assert ig.lm[it].iterations == info_o.lm[it].iterations, (k, it)
            assert ig.lm[it].termination == info_o.lm[it].termination, (k, it)
    assert_edges_equal(g.get_edges(), orc.extract(po, scans[K - 1], H, W))
    g.close()
    return modes, worst_t, worst_r


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["hdl64", "vlp16", "ouster128"])
def test_chain_mode_replay_against_the_oracle(orc, synth, monkeypatch, shape):
    for name in ("LIODOM_SPECULATE", "LIODOM_CHAIN", "LIODOM_KNN_OVERLAP", "LIODOM_SAFE_MODE", "LIODOM_PIPE_FLAGS"):
        monkeypatch.delenv(name, raising=False)
    modes, wt, wr = _chain_replay_against_the_oracle(orc, synth, shape, 12, data_stream=7)
    assert modes["chain"] == "1" and modes["speculate"] == "1" and modes["safe_mode"] == "0", modes
    assert wt < 1e-6 and wr < 1e-6


@pytest.mark.gpu
def test_chain_mode_replay_against_the_oracle_every_hand_over_wrong(orc, synth, monkeypatch):
    # LIODOM_SPECULATE=2: the iterate leaves as early as possible, i.e. practically always unconfirmed — every second pass,
    # every APPEND and every first pass of the next scan is repeated by the repair launches (k_knn_redo, k_chain_redo0)
    monkeypatch.setenv("LIODOM_SPECULATE", "2")
    modes, wt, wr = _chain_replay_against_the_oracle(orc, synth, "hdl64", 12, data_stream=8)
    assert modes["chain"] == "1" and modes["speculate"] == "2", modes
    assert wt < 1e-6 and wr < 1e-6


def _batch_run(orc, synth, scans, H, W, R, epr, P, S, env, monkeypatch, check_oracle=False):
    """One replay of `scans` (per data stream d: scans[d][k]) on an S-stream handle under `env`; returns per scan (poses, matches,
    correspondences of both passes for streams 0 .. D-1)."""
    for name in ("LIODOM_KNN8", "LIODOM_KNN_EXACT_ONLY", "LIODOM_KNN_SAVE", "LIODOM_HASH_INCR", "LIODOM_HB_SLACK", "LIODOM_HB_NEW_ROOM"):
        monkeypatch.delenv(name, raising=False)
    for name, val in env.items():
        monkeypatch.setenv(name, val)
    D, K = len(scans), len(scans[0])
    po, g = mk(orc, H, W, 0, R, epr, P, S=S, debug=1, pose_log_capacity=K + 8)
    modes = g.modes()
    g.alloc_resident(K)
    for s in range(S):
        for k in range(K):
            g.upload_scan(s, k, scans[s % D][k])
    out = []
    for k in range(K):
        maps = [g.local_map(d)[0] for d in range(D)] if check_oracle else None
        poses, infos = g.process_resident(k, H * W, H, W, readback=True, next_slot=(k + 1 if k + 1 < K else -1))
        assert all(i.status == 0 for i in infos), k
        corr = [[tuple(a.copy() for a in g.correspondences(it, stream=d)) for it in (0, 1)] for d in range(D)]
        if check_oracle and k > 0:
            for d in range(D):
                for it in (0, 1):
                    vk, ak, bk = orc.match_edges(po, maps[d], g.knn_queries(it, stream=d))
                    vg, ag, bg = corr[d][it]
                    assert np.array_equal(vk, vg) and np.array_equal(ak, ag) and np.array_equal(bk, bg), (k, d, it, np.nonzero((vk != vg) | (ak != ag) | (bk != bg))[0][:10])
        out.append((poses.copy(), [tuple(i.matches) for i in infos], corr))
    g.close()
    return modes, out


def _assert_batch_runs_equal(a, b, what):
    for k, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x[0].view(np.uint64), y[0].view(np.uint64)), (what, k)
        assert x[1] == y[1], (what, k)
        for cx, cy in zip(x[2], y[2]):
            for it in (0, 1):
                assert all(np.array_equal(p, q) for p, q in zip(cx[it], cy[it])), (what, k, it)


def test_knn8_equals_the_half_wave_kernel_and_its_own_modes(orc, synth, monkeypatch):
    """Lock-step batches search with k_knn8 (eight lanes per query: Best3 candidates per lane, neighbour cells probed lazily, 24
    candidates kept for the second pass — kernels_knn8.h).  Against k_knn<128> (LIODOM_KNN8=0: a half-wave per query) and against its
    own exact-list path (LIODOM_KNN_EXACT_ONLY=1), bound-only second pass (LIODOM_KNN_SAVE=1) and search-again second pass (=0):
    poses, match counts and the correspondences of both passes bit-identical, on a gentle and on a violent trajectory (many
    second-pass queries that the guard cannot certify).  Matches laser_odometry.cc:318-361."""
    H, W, R, epr, P, S, K = 16, 900, 6, 10, 5, 16, 12
    for yaw, speed in ((0.5, 0.1), (3.0, 0.6)):
        cfg = synth.make_cfg(H, W, 0, yaw_rate_deg=yaw, speed=speed)
        scans = [[synth.scan(cfg, 30 + d, k)[0] for k in range(K)] for d in range(3)]
        modes, base = _batch_run(orc, synth, scans, H, W, R, epr, P, S, {}, monkeypatch)
        assert modes["knn8"] == "1" and modes["line_gate_kernel"] == "1" and modes["hash_incr"] == "1"
        modes, old = _batch_run(orc, synth, scans, H, W, R, epr, P, S, {"LIODOM_KNN8": "0"}, monkeypatch)
        assert modes["knn8"] == "0"
        _assert_batch_runs_equal(base, old, ("half-wave kernel", yaw))
        assert modes["knn8"] == "0" and modes["hash_incr"] == "0"
        # LIODOM_HASH_INCR=0: the cell hash rebuilt from the whole window every scan (k_hash_build) instead of the new frame appended
        # to the table of the last rebuild, evicted frames' points skipped by their window index (k_hash_append): 12 scans at P = 5 —
        # three rebuild periods, seven evictions
        for env in ({"LIODOM_KNN_EXACT_ONLY": "1"}, {"LIODOM_KNN_SAVE": "1"}, {"LIODOM_KNN_SAVE": "0"}, {"LIODOM_HASH_INCR": "0"},
                    {"LIODOM_HASH_INCR": "0", "LIODOM_KNN_EXACT_ONLY": "1"},
                    {"LIODOM_HB_SLACK": "1", "LIODOM_HB_NEW_ROOM": "2"}):      # (cells run out of room almost every scan: the rebuild takes over)
            m2, other = _batch_run(orc, synth, scans, H, W, R, epr, P, S, env, monkeypatch)
            assert m2["hash_incr"] == ("0" if "LIODOM_HASH_INCR" in env else "1")
            _assert_batch_runs_equal(base, other, (env, yaw))
    assert sum(m[1] for m in base[-1][1]) > 100


def test_knn8_ties_are_ordered_by_window_index(orc, synth, monkeypatch):
    """test_knn_ties_are_ordered_by_window_index on the batch instance: a static sensor in a noise-free world, P = 2 — every
    query's neighbours come in pairs of equal float distance, FLANN's order (lower window index first) decides the line points;
    k_knn8 must send those queries through its exact lists.  Valid flags and both line-point indices equal the oracle's loop
    (laser_odometry.cc:320-361) on the GPU's own inputs."""
    H, W, R, epr, P, S, K = 16, 900, 6, 10, 2, 16, 6
    cfg = synth.make_cfg(H, W, 0, noise_sigma=0.0, yaw_rate_deg=0.0, speed=0.0)
    x = synth.scan(cfg, 2, 0)[0]
    modes, out = _batch_run(orc, synth, [[x] * K], H, W, R, epr, P, S, {}, monkeypatch, check_oracle=True)
    assert modes["knn8"] == "1"
    assert sum(int(c[0][1][0].sum()) for _, _, c in out[2:]) > 0      # accepted correspondences whose NN0 / NN1 are two copies of one point

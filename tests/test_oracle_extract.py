"""Oracle (oracle/liodom_oracle.cc) edge extraction vs the independent Python transcription
(tests/pyref.py) and hand-built known-answer cases.  CPU only.

The reference holds no tests or golden vectors and cannot be compiled here, so
these cross-checks — not reference outputs — are what pins the oracle ("parity unpinned").
"""
import numpy as np
import pytest

import pyref


def _cmp(orc, p, x, h, w, **kw):
    e = orc.extract(p, x, h, w)
    ref = pyref.extract(x, h, w, **kw)
    got = list(zip(e["ring"].tolist(), e["idx_in_ring"].tolist(), e["src"].tolist()))
    assert got == ref
    # edges carry the source point unchanged (feature_extractor.cc:275)
    assert np.array_equal(e["edges"].view(np.uint32), x[e["src"]].view(np.uint32))
    return e


def test_cfg1_vlp16_small(orc, synth):
    # BASELINE config 1: 16 x 900, R=6, epr=10
    cfg = synth.make_cfg(16, 900, 0)
    p = orc.make_params(scan_lines=16, scan_regions=6, edges_per_region=10, prev_frames=5)
    for k in (0, 3):
        x, _ = synth.scan(cfg, 0, k)
        e = _cmp(orc, p, x, 16, 900, lidar_type=0, scan_lines=16, scan_regions=6, edges_per_region=10)
        assert len(e["ring"]) > 100


def test_hdl64_narrow(orc, synth):
    cfg = synth.make_cfg(64, 300, 0)
    p = orc.make_params(scan_lines=64, scan_regions=8, edges_per_region=10)
    x, _ = synth.scan(cfg, 1, 2)
    _cmp(orc, p, x, 64, 300, lidar_type=0, scan_lines=64, scan_regions=8, edges_per_region=10)


def test_hdl32(orc, synth):
    cfg = synth.make_cfg(32, 400, 0)
    p = orc.make_params(scan_lines=32, scan_regions=4, edges_per_region=5)
    x, _ = synth.scan(cfg, 0, 1)
    offs, _ = orc.split(p, x, 32, 400)
    assert (np.diff(offs) > 0).sum() >= 20        # the synthetic elevations land in the 32-line bins
    _cmp(orc, p, x, 32, 400, lidar_type=0, scan_lines=32, scan_regions=4, edges_per_region=5)


def test_ouster_rows(orc, synth):
    cfg = synth.make_cfg(32, 512, 1)
    p = orc.make_params(lidar_type=1, scan_lines=32, scan_regions=8, edges_per_region=10)
    x, _ = synth.scan(cfg, 0, 0)
    _cmp(orc, p, x, 32, 512, lidar_type=1, scan_lines=32, scan_regions=8, edges_per_region=10)


def test_empty_and_invalid(orc):
    p = orc.make_params(scan_lines=16)
    e = orc.extract(p, np.zeros((0, 4), np.float32), 16, 0)
    assert len(e["ring"]) == 0
    x = np.full((16 * 50, 4), np.nan, np.float32)
    assert len(orc.extract(p, x, 16, 50)["ring"]) == 0
    # all points closer than min_range / farther than max_range are dropped (XY distance)
    x = np.zeros((100, 4), np.float32)
    x[:, 0] = 1.0
    x[:, 2] = 50.0
    offs, order = orc.split(p, x, 16, 0)
    assert len(order) == 0
    x[:, 0] = 80.0
    offs, order = orc.split(p, x, 16, 0)
    assert len(order) == 0


def test_range_boundaries_inclusive(orc):
    # comparisons are '>' and '<' (feature_extractor.cc:97): equality is valid
    p = orc.make_params(scan_lines=16, min_range=3.0, max_range=75.0)
    x = np.zeros((2, 4), np.float32)
    x[0, 0] = 3.0
    x[1, 0] = 75.0
    offs, order = orc.split(p, x, 16, 0)
    assert sorted(order.tolist()) == [0, 1]


def test_ring_formulas(orc):
    # spot values of the elevation -> ring maps (feature_extractor.cc:130-148), incl. truncation
    def ring_of(angle_deg, lines):
        p = orc.make_params(scan_lines=lines)
        d = 10.0
        x = np.array([[d, 0, d * np.tan(np.deg2rad(angle_deg)), 0]], np.float32)
        offs, order = orc.split(p, x, lines, 0)
        r = np.nonzero(np.diff(offs))[0]
        return int(r[0]) if len(r) else -1
    assert ring_of(1.95, 64) == 0
    assert ring_of(-8.4, 64) == 31
    assert ring_of(-8.78, 64) == 32
    assert ring_of(-24.28, 64) == 63
    assert ring_of(2.5, 64) == -1
    assert ring_of(-25.0, 64) == -1
    assert ring_of(-15.0, 16) == 0
    assert ring_of(15.0, 16) == 15
    assert ring_of(-17.0, 16) == 0        # int() truncates toward zero: (-2)/2+0.5 = -0.5 -> 0
    assert ring_of(-18.5, 16) == -1 or ring_of(-18.5, 16) == 0
    assert ring_of(17.0, 16) == -1
    assert ring_of(-30.0, 32) == 0
    assert ring_of(10.0, 32) == 30


def _jagged_ring(n, seed=0, amp=0.5):
    # one ring at elevation 0 for a 16-line sensor is ring 7/8; use -1 deg -> ring 7
    rng = np.random.default_rng(seed)
    phi = np.linspace(0, 2 * np.pi, n, endpoint=False)
    r = 20.0 + amp * ((np.arange(n) % 2) * 2 - 1) + 0.01 * rng.standard_normal(n)
    x = np.zeros((n, 4), np.float32)
    x[:, 0] = r * np.cos(phi)
    x[:, 1] = r * np.sin(phi)
    x[:, 2] = np.hypot(x[:, 0], x[:, 1]) * np.tan(np.deg2rad(-1.0))
    x[:, 3] = np.arange(n)
    return x


def test_epr_plus_one_picks_per_region(orc):
    # the stop test is picked_edges > epr, so a region yields epr+1 edges
    x = _jagged_ring(1800)
    p = orc.make_params(scan_lines=16, scan_regions=8, edges_per_region=10)
    e = orc.extract(p, x, 16, 0)
    assert len(e["ring"]) == 8 * 11
    total, sector = 1790, 1790 // 8
    reg = np.minimum((e["idx_in_ring"] - 5) // sector, 7)
    assert np.array_equal(np.bincount(reg, minlength=8), np.full(8, 11))
    ref = pyref.extract(x, 16, 0, lidar_type=0, scan_lines=16, scan_regions=8, edges_per_region=10)
    assert [t[1] for t in ref] == e["idx_in_ring"].tolist()


def test_suppression_spills_into_next_region(orc):
    # picked_ persists across the regions of a ring.  Smooth arc with two
    # spikes straddling a region boundary: the second (weaker) spike sits within 5 samples of
    # the first and must be suppressed although it belongs to the next region.
    n = 910                       # total = 900, 2 regions of 450: boundary between idx 454 | 455
    phi = np.linspace(0, 0.5, n)
    r = np.full(n, 20.0)
    r[453] += 0.08                # region 0 (trimmed index 448), strong
    r[456] += 0.05                # region 1 (trimmed index 451), weaker, 3 samples away
    x = np.zeros((n, 4), np.float32)
    x[:, 0] = r * np.cos(phi)
    x[:, 1] = r * np.sin(phi)
    x[:, 2] = -0.3
    p = orc.make_params(scan_lines=16, scan_regions=2, edges_per_region=3)
    e = orc.extract(p, x, 16, 0)
    ref = pyref.extract(x, 16, 0, lidar_type=0, scan_lines=16, scan_regions=2, edges_per_region=3)
    assert [t[1] for t in ref] == e["idx_in_ring"].tolist()
    idx = e["idx_in_ring"].tolist()
    assert 453 in idx
    # nothing within 5 samples after 453 may be picked, even though 455.. is a different region
    assert not any(453 < i <= 458 for i in idx)
    # control: without the strong spike the weaker one is picked
    r2 = np.full(n, 20.0)
    r2[456] += 0.05
    x2 = x.copy()
    x2[:, 0] = r2 * np.cos(phi)
    x2[:, 1] = r2 * np.sin(phi)
    e2 = orc.extract(p, x2, 16, 0)
    assert any(453 < i <= 459 for i in e2["idx_in_ring"].tolist())


def test_short_ring_skipped(orc):
    # rings with fewer than R*epr+10 points are skipped (feature_extractor.cc:188, params.cc:63)
    p = orc.make_params(scan_lines=16, scan_regions=8, edges_per_region=10)
    x = _jagged_ring(89)
    assert len(orc.extract(p, x, 16, 0)["ring"]) == 0
    x = _jagged_ring(90)
    e = orc.extract(p, x, 16, 0)
    ref = pyref.extract(x, 16, 0, lidar_type=0, scan_lines=16, scan_regions=8, edges_per_region=10)
    assert [t[1] for t in ref] == e["idx_in_ring"].tolist()
    assert len(ref) > 0


def test_curvature_values(orc):
    # smoothness: the 11-tap stencil is a FLOAT expression (pcl::PointXYZI floats, `10 * x` is
    # int * float) evaluated left to right and only then widened; squares and their sum are double
    # (feature_extractor.cc:196-229).  A float64 evaluation of the same sum differs in the last bits.
    x = _jagged_ring(200, seed=3)
    p = orc.make_params(scan_lines=16, scan_regions=2, edges_per_region=2)
    e = orc.extract(p, x, 16, 0, want_curv=True)
    P = x[:, :3].astype(np.float32)
    ten = np.float32(10)
    n_diff64 = 0
    for j in range(5, 195):
        d = np.zeros(3)
        d64 = np.zeros(3)
        for ax in range(3):
            s = P[j - 5, ax]
            for k in (-4, -3, -2, -1):
                s = s + P[j + k, ax]
            s = s - ten * P[j, ax]
            for k in (1, 2, 3, 4, 5):
                s = s + P[j + k, ax]
            assert s.dtype == np.float32
            d[ax] = float(s)
            Q = P.astype(np.float64)
            d64[ax] = Q[j - 5:j, ax].sum() - 10 * Q[j, ax] + Q[j + 1:j + 6, ax].sum()
        assert e["curv"][j] == d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
        n_diff64 += e["curv"][j] != d64[0] * d64[0] + d64[1] * d64[1] + d64[2] * d64[2]
    assert n_diff64 > 0          # the distinction is observable on this ring
    assert np.isnan(e["curv"][:5]).all() and np.isnan(e["curv"][195:200]).all()

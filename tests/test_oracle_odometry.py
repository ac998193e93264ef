"""Oracle odometry pieces vs independent NumPy / SciPy computations.  CPU only.

Parity with the reference is unpinned for these (PCL / FLANN / Eigen / Ceres are absent and the
reference has no tests); what is checked here is that the restatement does what SURVEY.md
Appendix A says: exact float 5-NN, FP64 covariance eigenvalue gate, the Point2LineFactor
residual (include/liodom/factors.hpp:71-105) with its autodiff Jacobian, and a Ceres-style
trust-region LM that reaches the minimiser of the same Huber cost.
"""
import numpy as np
import pytest


def _np_knn5(m, q):
    # float32 arithmetic in FLANN's order: ((dx*dx) + dy*dy) + dz*dz
    out_i = np.zeros((q.shape[0], 5), np.int64)
    out_d = np.zeros((q.shape[0], 5), np.float32)
    for i in range(q.shape[0]):
        dx = (q[i, 0] - m[:, 0]).astype(np.float32)
        dy = (q[i, 1] - m[:, 1]).astype(np.float32)
        dz = (q[i, 2] - m[:, 2]).astype(np.float32)
        d = (dx * dx).astype(np.float32)
        d = (d + (dy * dy).astype(np.float32)).astype(np.float32)
        d = (d + (dz * dz).astype(np.float32)).astype(np.float32)
        o = np.lexsort((np.arange(len(d)), d))[:5]
        out_i[i], out_d[i] = o, d[o]
    return out_i, out_d


def test_knn5_brute_and_kdtree(orc):
    rng = np.random.default_rng(0)
    m = np.zeros((3000, 4), np.float32)
    m[:, :3] = rng.uniform(-20, 20, (3000, 3))
    m[10] = m[11]                       # exact duplicate -> tie broken by lower index
    q = np.zeros((200, 4), np.float32)
    q[:, :3] = rng.uniform(-20, 20, (200, 3))
    q[0, :3] = m[10, :3]
    ri, rd = _np_knn5(m, q)
    for mode in (0, 1):
        i, d = orc.knn5(m, q, mode)
        assert np.array_equal(i, ri)
        assert np.array_equal(d.view(np.uint32), rd.view(np.uint32))
    assert ri[0, 0] == 10 and ri[0, 1] == 11


def test_knn5_fewer_than_five(orc):
    m = np.zeros((3, 4), np.float32)
    m[:, 0] = [1, 2, 3]
    i, d = orc.knn5(m, np.zeros((1, 4), np.float32), 0)
    assert i[0].tolist() == [0, 1, 2, -1, -1]


def test_eig3_vs_numpy(orc):
    rng = np.random.default_rng(1)
    for _ in range(200):
        pts = rng.normal(size=(5, 3)) * rng.uniform(0.01, 2, 3)
        c = pts - pts.mean(0)
        A = c.T @ c
        ev = orc.eig3([A[0, 0], A[0, 1], A[0, 2], A[1, 1], A[1, 2], A[2, 2]])
        ref = np.linalg.eigvalsh(A)
        assert np.allclose(ev, ref, rtol=1e-11, atol=1e-13 * ref[2])
    assert orc.eig3([0, 0, 0, 0, 0, 0]).tolist() == [0, 0, 0]
    assert np.allclose(orc.eig3([3, 0, 0, 1, 0, 2]), [1, 2, 3])
    # collinear points: two (near-)zero eigenvalues, gate lambda2 > 3*lambda1 holds
    pts = np.outer(np.arange(5.0), [0.3, 0.1, 1.0])
    c = pts - pts.mean(0)
    A = c.T @ c
    ev = orc.eig3([A[0, 0], A[0, 1], A[0, 2], A[1, 1], A[1, 2], A[2, 2]])
    assert ev[2] > 3 * ev[1] and abs(ev[1]) < 1e-12


def _rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _residual_np(q, t, p, a, b, mn=3.0, mx=75.0):
    lp = _rot(q) @ p + t
    nu = np.cross(lp - a, lp - b)
    de = a - b
    rho = np.hypot(p[0] - t[0], p[1] - t[1])
    w = 1.01 - (rho - mn) / (mx - mn)
    return w * nu / np.linalg.norm(de)


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def test_point2line_residual_and_jacobian(orc):
    rng = np.random.default_rng(2)
    for trial in range(50):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        if trial == 0:
            q = np.array([0.0, 0.0, 0.0, 1.0])       # slerp branch |d| >= 1-eps
        if trial == 1:
            q = -q                                    # w < 0 branch (scale1 negated)
        t = rng.normal(size=3) * 5
        p = rng.normal(size=3) * 20
        a = rng.normal(size=3) * 20
        b = a + rng.normal(size=3)
        r, J, Jg = orc.point2line(q, t, p, a, b)
        assert np.allclose(r, _residual_np(q, t, p, a, b), rtol=1e-10, atol=1e-10)
        # finite differences in the tangent space of EigenQuaternionParameterization
        h = 1e-6
        Jfd = np.zeros((3, 6))
        for k in range(6):
            d = np.zeros(6)
            d[k] = h
            rp = _residual_np(orc.quat_plus(q, d[:3]), t + d[3:], p, a, b)
            rm = _residual_np(orc.quat_plus(q, -d[:3]), t - d[3:], p, a, b)
            Jfd[:, k] = (rp - rm) / (2 * h)
        assert np.allclose(J, Jfd, rtol=1e-5, atol=1e-5 * max(1.0, np.abs(J).max()))
        # closed form of SURVEY.md A.4
        Rp = _rot(q) @ p
        de = a - b
        L = np.linalg.norm(de)
        rho = np.hypot(p[0] - t[0], p[1] - t[1])
        w = 1.01 - (rho - 3.0) / 72.0
        nu = np.cross(Rp + t - a, Rp + t - b)
        Jq = (2 * w / L) * _skew(de) @ _skew(Rp)
        dw = np.array([p[0] - t[0], p[1] - t[1], 0.0]) / (rho * 72.0)
        Jt = -(w / L) * _skew(de) + np.outer(nu / L, dw)
        assert np.allclose(J[:, :3], Jq, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(Jq).max()))
        assert np.allclose(J[:, 3:], Jt, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(Jt).max()))


def _make_problem(rng, n, noise=0.02, outliers=0):
    # vertical and horizontal line features observed from a displaced pose
    q_true = np.array([0.01, -0.02, 0.05, 1.0])
    q_true /= np.linalg.norm(q_true)
    t_true = np.array([0.3, -0.2, 0.05])
    R = _rot(q_true)
    blocks = []
    for i in range(n):
        base = rng.uniform(-30, 30, 3)
        base[2] = rng.uniform(-1.5, 4)
        if np.hypot(base[0], base[1]) < 4:
            base[0] += 8
        direction = np.array([0, 0, 1.0]) if i % 3 else rng.normal(size=3)
        direction /= np.linalg.norm(direction)
        a = base + direction * rng.uniform(0.05, 0.3)
        b = base - direction * rng.uniform(0.05, 0.3)
        pw = base + direction * rng.uniform(-0.5, 0.5) + rng.normal(size=3) * noise
        if i < outliers:
            pw += rng.normal(size=3) * 2.0
        p = R.T @ (pw - t_true)
        blocks.append(np.concatenate([p, a, b]))
    return np.array(blocks), q_true, t_true


def test_lm_decreases_cost_and_converges_to_scipy_minimum(orc):
    from scipy.optimize import minimize
    rng = np.random.default_rng(3)
    blocks, q_true, t_true = _make_problem(rng, 300, outliers=20)
    q0 = np.array([0.0, 0.0, 0.0, 1.0])
    t0 = np.zeros(3)
    c0 = orc.cost(blocks, q0, t0)
    q, t, tr = orc.lm_solve(blocks, q0, t0)
    assert tr.iterations <= 4 and tr.accepted >= 1
    assert abs(tr.initial_cost - c0) < 1e-9 * c0
    assert tr.final_cost < 0.2 * c0
    assert abs(orc.cost(blocks, q, t) - tr.final_cost) < 1e-9 * max(tr.final_cost, 1e-12)
    # iterate the 4-iteration solve to convergence and compare with a generic minimiser
    for _ in range(10):
        q, t, tr = orc.lm_solve(blocks, q, t)
    def f(v):
        qq = orc.quat_plus(q, v[:3])
        return orc.cost(blocks, qq, t + v[3:])
    res = minimize(f, np.zeros(6), method="Nelder-Mead", options=dict(xatol=1e-10, fatol=1e-16, maxiter=20000))
    assert res.fun >= orc.cost(blocks, q, t) * (1 - 1e-6)
    assert np.linalg.norm(res.x) < 1e-4
    assert np.linalg.norm(t - t_true) < 0.05


def test_lm_no_blocks_and_costs(orc):
    q, t, tr = orc.lm_solve(np.zeros((0, 9)), [0, 0, 0, 1.0], [1.0, 2, 3])
    assert tr.termination == 4 and q.tolist() == [0, 0, 0, 1] and t.tolist() == [1, 2, 3]
    # Huber(0.2): one block with |r| = 1 -> 0.5*(2*0.2*1 - 0.04) = 0.18
    p = np.array([10.0, 0, 0])
    a = np.array([10.0, 1.0, -1])
    b = np.array([10.0, 1.0, 1])
    blk = np.concatenate([p, a, b])[None]
    r = _residual_np([0, 0, 0, 1.0], np.zeros(3), p, a, b)
    s = r @ r
    expect = 0.5 * (2 * 0.2 * np.sqrt(s) - 0.04) if s > 0.04 else 0.5 * s
    assert abs(orc.cost(blk, [0, 0, 0, 1.0], [0, 0, 0.0]) - expect) < 1e-14


def test_transform_and_pose_ops(orc):
    rng = np.random.default_rng(4)
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    t = rng.normal(size=3)
    T, qb = orc.pose_ops(q, t)
    assert np.allclose(T[:, :3], _rot(q), atol=1e-15)
    assert np.allclose(qb, q, atol=1e-14) or np.allclose(qb, -q, atol=1e-14)
    x = np.zeros((100, 4), np.float32)
    x[:, :3] = rng.normal(size=(100, 3)) * 30
    x[:, 3] = np.arange(100)
    y = orc.transform(T, x)
    P = x[:, :3].astype(np.float64)
    ref = np.stack([((T[i, 0] * P[:, 0] + T[i, 1] * P[:, 1]) + T[i, 2] * P[:, 2]) + T[i, 3] for i in range(3)], 1)
    assert np.array_equal(y[:, :3], ref.astype(np.float32))
    assert np.array_equal(y[:, 3], x[:, 3])


def test_voxel_grid_properties(orc):
    rng = np.random.default_rng(5)
    x = np.zeros((5000, 4), np.float32)
    x[:, :3] = rng.uniform(-5, 5, (5000, 3))
    x[:, 3] = 1.0
    y = orc.voxel_grid(x, 0.4)
    cells = np.floor(x[:, :3] / np.float32(0.4)).astype(np.int64)
    assert len(y) == len(np.unique(cells, axis=0))
    ycells = np.floor(y[:, :3] / np.float32(0.4)).astype(np.int64)
    assert len(np.unique(ycells, axis=0)) == len(y)          # one centroid per leaf
    assert np.allclose(y[:, 3], 1.0)
    assert len(orc.voxel_grid(y, 0.4)) == len(y)              # idempotent in count


def test_window_fifo(orc):
    # LocalMapManager keeps the last P frames (laser_odometry.cc:34-60)
    p = orc.make_params(scan_lines=16, prev_frames=3)
    od = orc.Odometer(p)
    sizes = [7, 5, 9, 4, 6]
    for k, n in enumerate(sizes):
        e = np.zeros((n, 4), np.float32)
        e[:, 0] = 100 * (k + 1) + np.arange(n)          # far apart: no matches, pose stays identity
        e[:, 3] = k
        pose, info = od.step(e)
        assert np.allclose(pose, [0, 0, 0, 1, 0, 0, 0])
        assert od.window_frames() == min(k + 1, 3)
    w = od.window()
    assert len(w) == 9 + 4 + 6
    assert w[:, 3].tolist() == [2] * 9 + [3] * 4 + [4] * 6


def test_odometer_tracks_synthetic_stream(orc, synth):
    # BASELINE config 1 (16 x 900, R=6, epr=10, P=5): the estimated trajectory follows the
    # ground truth of the generator to ~10 cm over 12 scans (16 rings give few edges).
    cfg = synth.make_cfg(16, 900, 0)
    p = orc.make_params(scan_lines=16, scan_regions=6, edges_per_region=10, prev_frames=5)
    od = orc.Odometer(p)
    od_kd = orc.Odometer(orc.make_params(scan_lines=16, scan_regions=6, edges_per_region=10, prev_frames=5, knn_mode=1))
    for k in range(12):
        x, gt = synth.scan(cfg, 0, k)
        e = orc.extract(p, x, 16, 900)
        pose, info = od.step(e["edges"])
        pose_kd, _ = od_kd.step(e["edges"])
        assert np.array_equal(pose, pose_kd)          # kd-tree and brute force agree exactly
        if k > 0:
            assert info.matches[0] > 30 and info.lm[0].iterations >= 1
        # z is weakly observable from (mostly vertical) edge lines with only 16 rings
        assert np.linalg.norm(pose[4:6] - gt[4:6]) < 0.15 and abs(pose[6] - gt[6]) < 0.15
        dq = min(np.linalg.norm(pose[:4] - gt[:4]), np.linalg.norm(pose[:4] + gt[:4]))
        assert dq < 0.01


def test_imu_roll_pitch_override_matches_fixed_axis_euler(orc):
    """laser_odometry.cc:152-183 through the restated tf::Matrix3x3 getRPY / setRPY / getRotation:
    cross-check against scipy's extrinsic x-y-z (= fixed-axis roll, pitch, yaw) conversion."""
    from scipy.spatial.transform import Rotation as Rsc
    rng = np.random.default_rng(3)
    for trial in range(200):
        rpy_odom = rng.uniform([-0.5, -0.5, -3.1], [0.5, 0.5, 3.1])
        rpy_imu = rng.uniform([-0.6, -0.6, -3.1], [0.6, 0.6, 3.1])
        T = np.zeros((3, 4))
        T[:, :3] = Rsc.from_euler("xyz", rpy_odom).as_matrix()
        T[:, 3] = rng.uniform(-50, 50, 3)
        q_imu = Rsc.from_euler("xyz", rpy_imu).as_quat()          # [x y z w]
        out = orc.imu_override(T, q_imu)
        want = Rsc.from_euler("xyz", [rpy_imu[0], rpy_imu[1], rpy_odom[2]]).as_matrix()
        assert np.allclose(out[:, :3], want, atol=1e-12), trial
        assert np.array_equal(out[:, 3], T[:, 3])
        # with a laser -> base_link mounting transform the override acts in the base_link frame
        L = np.zeros((3, 4))
        L[:, :3] = Rsc.from_euler("xyz", rng.uniform(-0.3, 0.3, 3)).as_matrix()
        L[:, 3] = rng.uniform(-1, 1, 3)
        out2 = orc.imu_override(T, q_imu, L)
        T4, L4 = np.vstack([T, [0, 0, 0, 1]]), np.vstack([L, [0, 0, 0, 1]])
        bl = T4 @ L4
        yaw_bl = Rsc.from_matrix(bl[:3, :3]).as_euler("xyz")[2]
        bl[:3, :3] = Rsc.from_euler("xyz", [rpy_imu[0], rpy_imu[1], yaw_bl]).as_matrix()
        assert np.allclose(out2, (bl @ np.linalg.inv(L4))[:3], atol=1e-10), trial


def test_odometer_with_imu_keeps_roll_and_pitch(orc, synth):
    """use_imu: the prediction's roll / pitch are replaced by the IMU's before the solve; with a
    perfect IMU the trajectory stays as accurate as without."""
    H, W = 16, 900
    cfg = synth.make_cfg(H, W, 0)
    po = orc.make_params(scan_lines=H, scan_regions=6, edges_per_region=10, prev_frames=5, knn_mode=1)
    od = orc.Odometer(po)
    gt0 = None
    for k in range(8):
        x, gt = synth.scan(cfg, 0, k)
        if gt0 is None:
            gt0 = gt
        od.set_imu(gt[:4])
        pose, info = od.step(orc.extract(po, x, H, W)["edges"])
    assert info.matches[1] > 30
    assert np.linalg.norm(pose[4:] - (gt[4:] - gt0[4:])) < 0.2


def test_rotation_mode_soak(orc, synth):
    """Eigen::Transform::rotation() (laser_odometry.cc:186): with Eigen 3.3's polar factor
    (pose_rotation_mode 1, the default) the pose recursion T <- T (T_prev^-1 T), quaternion <- matrix
    <- quaternion stays normalised through any accumulated rotation; with the Eigen >= 3.4 alias of
    linear() (mode 0) the quaternion norm error grows once the sensor has turned ~90 deg and tracking
    is lost.  1 deg/scan, 16 x 900, 150 scans."""
    H, W, R, epr, P = 16, 900, 6, 10, 5
    cfg = synth.make_cfg(H, W, 0, yaw_rate_deg=1.0)
    res = {}
    for mode in (0, 1):
        po = orc.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, knn_mode=1,
                             pose_rotation_mode=mode)
        od = orc.Odometer(po)
        errs, qn = [], []
        for k in range(150):
            x, gt = synth.scan(cfg, 0, k)
            pose, _ = od.step(orc.extract(po, x, H, W)["edges"])
            errs.append(np.linalg.norm(pose[4:] - gt[4:]))
            qn.append(abs(np.linalg.norm(pose[:4]) - 1.0))
        od.close()
        res[mode] = (np.array(errs), np.array(qn))
    # both modes agree while the accumulated rotation is small ...
    assert np.allclose(res[0][0][:60], res[1][0][:60], atol=1e-6)
    # ... the polar mode keeps a unit quaternion and keeps tracking; the alias mode does neither
    assert res[1][1].max() < 1e-14 and res[1][0].max() < 1.0
    assert res[0][1][100:].max() > 1e-8 and res[0][0][-1] > 10.0


def test_match_edges_equals_the_odometers_own_loop(orc, synth):
    """orc.match_edges (addEdgeConstraints' per-edge loop on explicit inputs, used by the GPU tests to check the
    kNN + line-gate kernel on identical inputs) against the correspondences the Odometer recorded itself."""
    H, W = 16, 900
    cfg = synth.make_cfg(H, W, 0)
    po = orc.make_params(scan_lines=H, scan_regions=6, edges_per_region=10, prev_frames=5, knn_mode=1)
    od = orc.Odometer(po)
    for k in range(4):
        x, _ = synth.scan(cfg, 0, k)
        e = orc.extract(po, x, H, W)
        win = od.window()
        od.step(e["edges"])
        if k == 0:
            continue
        for it in (0, 1):
            v, a, b = od.last_corr(it)
            q = od.last_queries(it)
            v2, a2, b2 = orc.match_edges(po, win, q)
            assert np.array_equal(v, v2) and np.array_equal(a, a2) and np.array_equal(b, b2)
            assert v.sum() > 10
    od.close()


def test_parameter_tolerance_decision_is_the_ceres_form():
    """lm_update (liodom_math.h) decides the parameter tolerance on squares wherever that cannot differ from Ceres' form
    sqrt(step_sq) <= ptol, ptol = 1e-8 (|x| + 1e-8) — outside a relative band of 1e-12 around ptol^2 — and evaluates Ceres' form itself
    inside the band.  Checked here in NumPy: (a) the two forms can differ only within a few ulp of the boundary,
    (b) never outside the band, (c) the banded decision equals the literal one everywhere."""
    rng = np.random.default_rng(11)
    x_norm = rng.uniform(0.5, 200.0, size=200000)
    ptol = 1e-8 * (x_norm + 1e-8)
    k = rng.integers(-64, 65, size=x_norm.size)          # step norms within +-64 ulp of the boundary
    step = ptol * (1.0 + k * np.finfo(np.float64).eps)
    broad = ptol * rng.uniform(0.0, 3.0, size=x_norm.size)
    for st in (step, broad):
        step_sq = st * st
        lit = np.sqrt(step_sq) <= ptol
        p2 = ptol * ptol
        banded = step_sq <= p2 * (1.0 - 1e-12)
        inside = ~banded & (step_sq <= p2 * (1.0 + 1e-12))
        banded = banded | (inside & lit)
        assert np.array_equal(banded, lit)
        squares = step_sq <= p2
        assert np.array_equal(squares[~inside & ~(step_sq <= p2 * (1.0 - 1e-12))], lit[~inside & ~(step_sq <= p2 * (1.0 - 1e-12))])
    differ = (np.sqrt(step * step) <= ptol) != (step * step <= ptol * ptol)
    assert np.all(np.abs(k[differ]) <= 4)      # (a): wherever squares alone differ from Ceres' decision, it is on the knife edge
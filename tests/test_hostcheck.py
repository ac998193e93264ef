"""Product arithmetic (liodom_amd/csrc/liodom_math.h, compiled for the host by tests/hostcheck.cc)
vs the oracle.  CPU only — this is how the analytic Jacobian, the normal-equation accumulator and
the LM controller that run inside the HIP kernels are checked before any GPU is involved."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from test_oracle_odometry import _make_problem

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hc():
    so = os.path.join(HERE, "libhostcheck.so")
    src = os.path.join(HERE, "hostcheck.cc")
    hdr = os.path.join(HERE, "..", "liodom_amd", "csrc", "liodom_math.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", so, src])
    L = C.CDLL(so)
    dp = C.POINTER(C.c_double)
    fp = C.POINTER(C.c_float)
    L.hc_velodyne_ring.restype = C.c_int
    L.hc_velodyne_ring.argtypes = [C.c_double] * 5 + [C.c_int]
    L.hc_curvature.restype = C.c_double
    L.hc_curvature.argtypes = [fp, fp, fp, C.c_int]
    L.hc_eig3.argtypes = [dp, dp]
    L.hc_line_gate.restype = C.c_int
    L.hc_line_gate.argtypes = [fp, fp, fp]
    L.hc_transform.argtypes = [dp, fp, C.c_int, fp]
    L.hc_pose_ops.argtypes = [dp, dp, dp, dp]
    L.hc_predict.argtypes = [dp, dp, dp]
    L.hc_imu_override.argtypes = [dp, dp, dp, dp, C.c_int]
    L.hc_odom_message.argtypes = [dp, dp, dp, C.c_double, dp, C.c_int]
    L.hc_rotation_of.argtypes = [dp, C.c_int, dp]
    L.hc_lm_controller.restype = C.c_int
    L.hc_lm_controller.argtypes = [dp, C.c_int, dp, dp, dp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.hc_accumulate.argtypes = [dp, C.c_int, dp, dp, C.c_double, C.c_double, dp]
    L.hc_lm_solve.restype = C.c_int
    L.hc_lm_solve.argtypes = [dp, C.c_int, dp, dp, C.c_double, C.c_double, C.c_int,
                              C.POINTER(C.c_int), C.POINTER(C.c_int), dp, dp]
    L.hc_lm_ptol_decision.restype = C.c_int
    L.hc_lm_ptol_decision.argtypes = [dp, dp, dp, dp, C.c_double]
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def test_ring_and_curvature_bitwise(hc, orc, synth):
    cfg = synth.make_cfg(64, 200, 0)
    x, _ = synth.scan(cfg, 0, 0)
    p = orc.make_params(scan_lines=64)
    offs, order = orc.split(p, x, 64, 200)
    ring_of = np.full(len(x), -1)
    for r in range(64):
        ring_of[order[offs[r]:offs[r + 1]]] = r
    for i in range(0, len(x), 7):
        got = hc.hc_velodyne_ring(float(x[i, 0]), float(x[i, 1]), float(x[i, 2]), 3.0, 75.0, 64)
        assert (got if got >= 0 else -1) == ring_of[i]
    e = orc.extract(p, x, 64, 200, want_curv=True)
    r = 40
    pts = np.ascontiguousarray(x[order[offs[r]:offs[r + 1]]])
    px, py, pz = [np.ascontiguousarray(pts[:, k]) for k in range(3)]
    for j in range(5, len(pts) - 5, 3):
        assert hc.hc_curvature(_fp(px), _fp(py), _fp(pz), j) == e["curv"][offs[r] + j]


def test_eig_and_gate_bitwise(hc, orc):
    rng = np.random.default_rng(0)
    for _ in range(300):
        pts = (rng.normal(size=(5, 3)) * rng.uniform(0.001, 1, 3)).astype(np.float32)
        c = pts.astype(np.float64)
        cen = np.zeros(3)
        for j in range(5):
            cen = cen + c[j]
        cen = cen / 5.0
        cov = np.zeros(6)
        for j in range(5):
            z = c[j] - cen
            cov = cov + np.array([z[0] * z[0], z[0] * z[1], z[0] * z[2], z[1] * z[1], z[1] * z[2], z[2] * z[2]])
        ev_o = orc.eig3(cov)
        ev_h = np.zeros(3)
        hc.hc_eig3(_dp(cov), _dp(ev_h))
        assert np.array_equal(ev_o, ev_h)
        nx, ny, nz = [np.ascontiguousarray(pts[:, k]) for k in range(3)]
        assert hc.hc_line_gate(_fp(nx), _fp(ny), _fp(nz)) == int(ev_o[2] > 3 * ev_o[1])


def test_line_gate_decision_near_the_threshold(hc, orc):
    """The two-stage closed-form gate (float trigonometry, then double, then the oracle's iterative solver) must
    take the oracle's decision lambda_max > 3 lambda_mid everywhere, also when the ratio is within 1e-3 ... 1e-9
    of 3 (where the cheaper stages must hand over) and for degenerate neighbourhoods."""
    rng = np.random.default_rng(42)
    n_near = 0
    for trial in range(6000):
        jitter = 10.0 ** rng.uniform(-10, -2)
        ratio = 3.0 * (1.0 + rng.choice([-1.0, 1.0]) * jitter) if trial % 2 else rng.uniform(1.0, 9.0)
        # five points with a scatter matrix of eigenvalues ~ (ratio, 1, eps) in a random frame
        base = rng.normal(size=(5, 3))
        base -= base.mean(0)
        u, sv, vt = np.linalg.svd(base, full_matrices=False)
        target = np.sqrt(np.array([ratio, 1.0, rng.uniform(0, 0.5)]))
        pts = (u * target) @ vt * rng.uniform(0.01, 0.5) + rng.uniform(-50, 50, 3)
        if trial % 50 == 0:
            pts[:] = pts[0]                                   # degenerate: five coincident points
        pts = pts.astype(np.float32)
        c = pts.astype(np.float64)
        cen = np.zeros(3)
        for j in range(5):
            cen = cen + c[j]
        cen = cen / 5.0
        cov = np.zeros(6)
        for j in range(5):
            z = c[j] - cen
            cov = cov + np.array([z[0] * z[0], z[0] * z[1], z[0] * z[2], z[1] * z[1], z[1] * z[2], z[2] * z[2]])
        ev_o = orc.eig3(cov)
        nx, ny, nz = [np.ascontiguousarray(pts[:, k]) for k in range(3)]
        assert hc.hc_line_gate(_fp(nx), _fp(ny), _fp(nz)) == int(ev_o[2] > 3 * ev_o[1]), (trial, ev_o)
        n_near += abs(ev_o[2] - 3 * ev_o[1]) < 1e-4 * ev_o[2]
    assert n_near > 200          # the hand-over bands were exercised


def test_transform_pose_bitwise(hc, orc):
    rng = np.random.default_rng(1)
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    t = rng.normal(size=3) * 10
    T_o, qb_o = orc.pose_ops(q, t)
    T_h = np.zeros(12)
    qb_h = np.zeros(4)
    hc.hc_pose_ops(_dp(q), _dp(t), _dp(T_h), _dp(qb_h))
    assert np.array_equal(T_o.reshape(12), T_h) and np.array_equal(qb_o, qb_h)
    x = np.zeros((64, 4), np.float32)
    x[:, :3] = rng.normal(size=(64, 3)) * 40
    out = np.zeros_like(x)
    hc.hc_transform(_dp(T_h), _fp(x), 64, _fp(out))
    assert np.array_equal(out, orc.transform(T_o, x))


def test_accumulator_matches_oracle_jacobians(hc, orc):
    rng = np.random.default_rng(2)
    blocks, q_true, t_true = _make_problem(rng, 60, outliers=10)
    q = np.array([0.02, -0.01, 0.03, 1.0])
    q /= np.linalg.norm(q)
    t = np.array([0.1, -0.1, 0.0])
    acc = np.zeros(29)
    hc.hc_accumulate(_dp(blocks), len(blocks), _dp(q), _dp(t), 3.0, 75.0, _dp(acc))
    cost, g, Hm = 0.0, np.zeros(6), np.zeros((6, 6))
    for b in blocks:
        r, J, _ = orc.point2line(q, t, b[:3], b[3:6], b[6:9])
        s = r @ r
        rho1 = 1.0 if s <= 0.04 else 0.2 / np.sqrt(s)
        cost += 0.5 * (s if s <= 0.04 else 2 * 0.2 * np.sqrt(s) - 0.04)
        g += rho1 * (J.T @ r)
        Hm += rho1 * (J.T @ J)
    assert abs(acc[0] - cost) <= 1e-12 * cost
    assert np.allclose(acc[1:7], g, rtol=1e-10, atol=1e-10 * np.abs(g).max())
    Hacc = np.zeros((6, 6))
    k = 7
    for i in range(6):
        for j in range(i, 6):
            Hacc[i, j] = Hacc[j, i] = acc[k]
            k += 1
    assert np.allclose(Hacc, Hm, rtol=1e-10, atol=1e-10 * np.abs(Hm).max())
    assert acc[28] == 0


@pytest.mark.parametrize("seed,n,outliers", [(3, 300, 20), (4, 40, 0), (5, 1500, 200), (6, 12, 3)])
def test_lm_controller_matches_oracle(hc, orc, seed, n, outliers):
    # same iteration count, termination reason and accepted steps; pose equal to ~1e-10
    # (normal equations + Cholesky here vs Householder QR on the stacked Jacobian in the oracle)
    rng = np.random.default_rng(seed)
    blocks, _, _ = _make_problem(rng, n, outliers=outliers)
    for apply_ftol in (0, 1):
        q0 = np.array([0.0, 0.0, 0.0, 1.0])
        t0 = np.zeros(3)
        qo, to, tr = orc.lm_solve(blocks, q0, t0, apply_on_ftol=apply_ftol)
        qh, th = q0.copy(), t0.copy()
        it, acc_n = C.c_int(), C.c_int()
        ic, fc = C.c_double(), C.c_double()
        term = hc.hc_lm_solve(_dp(blocks), len(blocks), _dp(qh), _dp(th), 3.0, 75.0, apply_ftol,
                              C.byref(it), C.byref(acc_n), C.byref(ic), C.byref(fc))
        assert (term, it.value, acc_n.value) == (tr.termination, tr.iterations, tr.accepted)
        assert abs(ic.value - tr.initial_cost) <= 1e-12 * tr.initial_cost
        assert abs(fc.value - tr.final_cost) <= 1e-9 * max(tr.final_cost, 1e-12)
        assert np.allclose(qh, qo, atol=1e-9) and np.allclose(th, to, atol=1e-9)
        # a second solve from the result (the reference's second outer iteration)
        qo2, to2, tr2 = orc.lm_solve(blocks, qo, to, apply_on_ftol=apply_ftol)
        term2 = hc.hc_lm_solve(_dp(blocks), len(blocks), _dp(qh), _dp(th), 3.0, 75.0, apply_ftol,
                               C.byref(it), C.byref(acc_n), C.byref(ic), C.byref(fc))
        assert np.allclose(qh, qo2, atol=1e-8) and np.allclose(th, to2, atol=1e-8)


def test_lm_degenerate_inputs(hc, orc):
    q = np.array([0.0, 0, 0, 1.0])
    t = np.zeros(3)
    it, acc_n = C.c_int(), C.c_int()
    ic, fc = C.c_double(), C.c_double()
    term = hc.hc_lm_solve(_dp(np.zeros((1, 9))), 0, _dp(q), _dp(t), 3.0, 75.0, 0, C.byref(it), C.byref(acc_n), C.byref(ic), C.byref(fc))
    assert term == 4 and it.value == 0
    # a == b -> |de| = 0 -> non-finite residual: Ceres rejects the evaluation, pose unchanged
    blk = np.array([[10.0, 1, 0, 10, 1, 1, 10, 1, 1]])
    term = hc.hc_lm_solve(_dp(blk), 1, _dp(q), _dp(t), 3.0, 75.0, 0, C.byref(it), C.byref(acc_n), C.byref(ic), C.byref(fc))
    assert term == 5 and q.tolist() == [0, 0, 0, 1]
    qo, to, tr = orc.lm_solve(blk, [0, 0, 0, 1.0], [0, 0, 0.0])
    assert tr.termination == 5


def test_imu_override_and_odom_message(hc, orc):
    """use_imu override (laser_odometry.cc:152-183) and publishOdom's numbers (:395-436): the
    product's header against the oracle, plus scipy for the twist of a known motion."""
    from scipy.spatial.transform import Rotation as Rsc
    rng = np.random.default_rng(5)
    for trial in range(100):
        def rand_iso(ang, tr):
            T = np.zeros((3, 4))
            T[:, :3] = Rsc.from_euler("xyz", rng.uniform(-ang, ang, 3)).as_matrix()
            T[:, 3] = rng.uniform(-tr, tr, 3)
            return T
        T, L = rand_iso(3.0 if trial % 2 else 0.4, 40.0), rand_iso(0.3, 1.0)
        q = Rsc.from_euler("xyz", rng.uniform(-0.7, 0.7, 3)).as_quat()
        mode = trial % 2          # both Transform::rotation() semantics
        got = np.zeros(12)
        hc.hc_imu_override(_dp(T.reshape(12).copy()), _dp(q.copy()), _dp(L.reshape(12).copy()), _dp(got), mode)
        assert np.allclose(got.reshape(3, 4), orc.imu_override(T, q, L, rotation_mode=mode), atol=1e-13)
        # publishOdom: previous pose -> current pose = previous * known motion
        step = rand_iso(0.05, 0.3)
        T4, S4 = np.vstack([T, [0, 0, 0, 1]]), np.vstack([step, [0, 0, 0, 1]])
        cur = (T4 @ S4)[:3]
        msg = np.zeros(13)
        hc.hc_odom_message(_dp(T.reshape(12).copy()), _dp(cur.reshape(12).copy()), _dp(L.reshape(12).copy()), 0.1, _dp(msg), mode)
        assert np.allclose(msg, orc.publish_odom(T, cur, 0.1, L, rotation_mode=mode), atol=1e-12)
    # identity mounting: twist = motion / dt, orientation = pose quaternion
    I = np.eye(4)[:3]
    step = np.zeros((3, 4)); step[:, :3] = Rsc.from_euler("xyz", [0.01, -0.02, 0.03]).as_matrix(); step[:, 3] = [0.1, 0.02, -0.01]
    msg = orc.publish_odom(I, step, 0.1)
    assert np.allclose(msg[7:10], step[:, 3] / 0.1) and np.allclose(msg[10:13], np.array([0.01, -0.02, 0.03]) / 0.1, atol=1e-12)
    assert np.allclose(msg[:4], Rsc.from_matrix(step[:, :3]).as_quat(), atol=1e-12)


def test_rotation_of_is_the_polar_factor(hc, orc):
    """Eigen 3.3 Transform::rotation() = U V^T of linear() (computeRotationScaling).  The oracle gets it
    from a one-sided Jacobi SVD, the product from Newton's iteration; both against numpy's SVD, on
    slightly non-orthonormal poses (what toRotationMatrix of a non-unit quaternion produces) and on
    badly scaled ones.  Mode 0 (Eigen >= 3.4) returns linear() untouched."""
    from scipy.spatial.transform import Rotation as Rsc
    rng = np.random.default_rng(11)
    for trial in range(200):
        R = Rsc.from_euler("xyz", rng.uniform(-3, 3, 3)).as_matrix()
        eps = [1e-15, 1e-9, 1e-4, 0.3][trial % 4]
        A = R @ (np.eye(3) + eps * rng.standard_normal((3, 3)))
        if trial % 16 == 15:
            A *= rng.uniform(0.2, 5.0)
        T = np.zeros((3, 4)); T[:, :3] = A; T[:, 3] = rng.uniform(-50, 50, 3)
        U, _, Vt = np.linalg.svd(A)
        want = U @ Vt
        assert np.linalg.det(want) > 0
        o = orc.rotation_of(T, 1)
        g = np.zeros(12)
        hc.hc_rotation_of(_dp(T.reshape(12).copy()), 1, _dp(g))
        g = g.reshape(3, 4)
        assert np.allclose(o[:, :3], want, atol=5e-15) and np.allclose(g[:, :3], want, atol=5e-15)
        assert np.array_equal(o[:, 3], T[:, 3]) and np.array_equal(g[:, 3], T[:, 3])
        assert np.allclose(g[:, :3].T @ g[:, :3], np.eye(3), atol=5e-15)
        g0 = np.zeros(12)
        hc.hc_rotation_of(_dp(T.reshape(12).copy()), 0, _dp(g0))
        assert np.array_equal(g0.reshape(3, 4), T) and np.array_equal(orc.rotation_of(T, 0), T)


def test_invalid_step_halves_the_radius(hc, orc):
    """TrustRegionMinimizer::HandleInvalidStep -> LevenbergMarquardtStrategy::StepIsInvalid():
    radius *= 0.5 (not StepRejected's radius /= decrease_factor; decrease_factor *= 2), every invalid
    step counts as an iteration, 4 iterations (src/laser_odometry.cc:214) end the solve.  Forced with
    normal equations whose Cholesky fails (a non-finite diagonal entry): model_cost_change <= 0."""
    acc = np.zeros(29)
    acc[0] = 1.0                        # cost
    acc[1:7] = [1.0, -2.0, 0.5, 0.1, 0.2, -0.3]     # gradient (above the 1e-10 gradient tolerance)
    diag = [7, 13, 18, 22, 25, 27]
    for i, d in enumerate(diag):
        acc[d] = 4.0 + i
    acc[7 + 1] = np.inf                 # H01 = inf: pivot 1 of the Cholesky becomes -inf / NaN -> step invalid
    radius = np.zeros(16)
    it, calls = C.c_int(), C.c_int()
    q, t = np.array([0, 0, 0, 1.0]), np.zeros(3)
    term = hc.hc_lm_controller(_dp(acc), 3, _dp(q), _dp(t), _dp(radius), 16, C.byref(it), C.byref(calls))
    # four invalid steps inside lm_begin's first proposal loop: 1e4 -> 5e3 -> 2.5e3 -> 1.25e3 -> 625, then max-iter
    assert term == 0 and it.value == 4 and calls.value == 1
    assert radius[0] == 1e4 * 0.5 ** 4
    # the oracle's solver with its linear solver made to fail (fault injection): same radius sequence
    from test_oracle_odometry import _make_problem as mk
    blocks = mk(np.random.default_rng(3), 60)[0]
    q0, t0 = np.array([0, 0, 0, 1.0]), np.zeros(3)
    orc.lib().orc_debug_fail_linear_solves(4)
    qo, to, tr = orc.lm_solve(blocks, q0, t0)
    orc.lib().orc_debug_fail_linear_solves(0)
    assert tr.termination == 0 and tr.iterations == 4 and tr.accepted == 0
    assert [tr.radius[i] for i in range(1, 5)] == [1e4, 5e3, 2.5e3, 1.25e3]
    assert np.array_equal(qo, q0) and np.array_equal(to, t0)
    # one failure, then normal steps: the radius of the first real step is the halved one
    orc.lib().orc_debug_fail_linear_solves(1)
    qo, to, tr = orc.lm_solve(blocks, q0, t0)
    assert tr.radius[1] == 1e4 and tr.radius[2] == 5e3 and tr.iterations >= 2


def test_lm_update_on_the_parameter_tolerance_boundary(hc):
    """Ceres' TrustRegionMinimizer ends a solve when step_norm <= parameter_tolerance * (x_norm + parameter_tolerance), with
    step_norm = sqrt(step . step) (the oracle has that expression literally, oracle/liodom_oracle.cc lm_solve;
    src/laser_odometry.cc:212-218 keeps the default 1e-8).  The product's lm_update compares squares and evaluates the literal
    expression only inside a 1e-12 band around the boundary: driven ONTO the boundary (steps within +-40 ulp of ptol, and exactly
    on it), its decision must be the literal one every time."""
    eps = np.finfo(np.float64).eps
    rng = np.random.default_rng(5)
    n_hit = n_miss = 0
    for x_norm in list(rng.uniform(0.5, 300.0, size=300)) + [1.0, 2.0, 64.0]:
        ptol = 1e-8 * (x_norm + 1e-8)
        for k in list(range(-40, 41)) + [-10 ** 5, 10 ** 5, -10 ** 9, 10 ** 9]:
            step = ptol * (1.0 + k * eps)
            q = np.array([0.0, 0.0, 0.0, 1.0]); t = np.array([1.0, 2.0, 3.0])
            ct = t.copy(); ct[0] = t[0] - step          # dt = t - cand_t: one non-zero component, so step_sq = fl(d * d) in every build
            d = t[0] - ct[0]
            literal = np.sqrt(d * d) <= ptol
            term = hc.hc_lm_ptol_decision(_dp(q), _dp(t), _dp(q.copy()), _dp(ct), x_norm)
            assert term == (1 if literal else 2), "x_norm %r, %d ulp from the boundary: termination %d, Ceres form says %s" % (x_norm, k, term, literal)
            n_hit += int(literal); n_miss += int(not literal)
    assert n_hit > 5000 and n_miss > 5000

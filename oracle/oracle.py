"""ctypes binding of the CPU oracle (oracle/liodom_oracle.cc).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (liodom_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libliodom_oracle.so")


class OrcParams(C.Structure):
    # mirrors orc_params_t (numeric fields of liodom::Params, include/liodom/params.h:33-49)
    _fields_ = [
        ("min_range", C.c_double),
        ("max_range", C.c_double),
        ("lidar_type", C.c_int32),
        ("scan_lines", C.c_int32),
        ("scan_regions", C.c_int32),
        ("edges_per_region", C.c_int32),
        ("min_points_per_scan", C.c_int64),
        ("local_map_size", C.c_int64),
        ("filter_local_map", C.c_int32),
        ("mapping", C.c_int32),
        ("lm_apply_step_on_ftol", C.c_int32),
        ("knn_mode", C.c_int32),
        ("pose_rotation_mode", C.c_int32),
        ("pad_", C.c_int32),
    ]


class LmTrace(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32), ("accepted", C.c_int32), ("termination", C.c_int32), ("pad", C.c_int32),
        ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("cost", C.c_double * 5), ("radius", C.c_double * 5),
        ("step_ok", C.c_int32 * 5), ("pad2", C.c_int32),
    ]


class StepInfo(C.Structure):
    _fields_ = [("n_edges", C.c_int32), ("map_points", C.c_int32), ("matches", C.c_int32 * 2), ("lm", LmTrace * 2)]


def _src_hash():
    import hashlib
    h = hashlib.sha256()
    for name in ("liodom_oracle.cc", "Makefile"):
        with open(os.path.join(_HERE, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False):
    """(Re)builds the oracle library when its source hash differs from the one recorded at its last
    build (mtimes do not survive the copy to the GPU box)."""
    import fcntl
    os.makedirs(os.path.join(_HERE, "_build"), exist_ok=True)
    stamp = _LIB_PATH + ".srchash"
    with open(_LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        want = _src_hash()
        have = open(stamp).read().strip() if os.path.exists(stamp) else ""
        if force or not os.path.exists(_LIB_PATH) or have != want:
            subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
            with open(stamp, "w") as f:
                f.write(want + "\n")
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        L.orc_split.restype = C.c_int
        L.orc_split.argtypes = [C.POINTER(OrcParams), fp, C.c_int64, C.c_int, C.c_int, ip, ip]
        L.orc_extract.restype = C.c_int
        L.orc_extract.argtypes = [C.POINTER(OrcParams), fp, C.c_int64, C.c_int, C.c_int, fp, ip, ip, ip, C.c_int, dp]
        L.orc_odom_create.restype = C.c_void_p
        L.orc_odom_create.argtypes = [C.POINTER(OrcParams)]
        L.orc_odom_destroy.argtypes = [C.c_void_p]
        L.orc_odom_set_imu.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_odom_set_laser_to_base.argtypes = [C.c_void_p, dp]
        L.orc_imu_override.argtypes = [dp, dp, dp, dp, C.c_int]
        L.orc_publish_odom.argtypes = [dp, dp, dp, C.c_double, dp, C.c_int]
        L.orc_rotation_of.argtypes = [dp, C.c_int, dp]
        L.orc_set_threads.argtypes = [C.c_int, C.c_int]
        L.orc_odom_step.restype = C.c_int
        L.orc_odom_step.argtypes = [C.c_void_p, fp, C.c_int, dp, C.POINTER(StepInfo)]
        L.orc_odom_last_corr.restype = C.c_int
        L.orc_odom_last_corr.argtypes = [C.c_void_p, C.c_int, ip, ip, ip, C.c_int]
        L.orc_odom_last_queries.restype = C.c_int
        L.orc_odom_last_queries.argtypes = [C.c_void_p, C.c_int, fp, C.c_int]
        L.orc_match_edges.argtypes = [C.POINTER(OrcParams), fp, C.c_int64, fp, C.c_int64, ip, ip, ip]
        L.orc_odom_window_size.restype = C.c_int64
        L.orc_odom_window_size.argtypes = [C.c_void_p]
        L.orc_odom_window_frames.restype = C.c_int
        L.orc_odom_window_frames.argtypes = [C.c_void_p]
        L.orc_odom_get_window.restype = C.c_int64
        L.orc_odom_get_window.argtypes = [C.c_void_p, fp, C.c_int64]
        L.orc_odom_set_received_map.argtypes = [C.c_void_p, fp, C.c_int64]
        L.orc_odom_get_state.argtypes = [C.c_void_p, dp, dp]
        L.orc_odom_get_received_map.restype = C.c_int64
        L.orc_odom_get_received_map.argtypes = [C.c_void_p, fp, C.c_int64]
        L.orc_odom_map_total.restype = C.c_int64
        L.orc_odom_map_total.argtypes = [C.c_void_p]
        L.orc_map_create.restype = C.c_void_p
        L.orc_map_create.argtypes = [C.c_double, C.c_double, C.c_double]
        L.orc_map_destroy.argtypes = [C.c_void_p]
        L.orc_map_update.argtypes = [C.c_void_p, fp, C.c_int64, dp]
        L.orc_map_get_local.restype = C.c_int64
        L.orc_map_get_local.argtypes = [C.c_void_p, dp, C.c_int, C.c_int, fp, C.c_int64]
        L.orc_map_get_all.restype = C.c_int64
        L.orc_map_get_all.argtypes = [C.c_void_p, fp, C.c_int64]
        L.orc_map_num_cells.restype = C.c_int
        L.orc_map_num_cells.argtypes = [C.c_void_p]
        L.orc_knn5.argtypes = [fp, C.c_int64, fp, C.c_int64, C.c_int, ip, fp]
        L.orc_eig3.argtypes = [dp, dp]
        L.orc_point2line.restype = C.c_int
        L.orc_point2line.argtypes = [dp, dp, dp, dp, dp, C.c_double, C.c_double, dp, dp, dp]
        L.orc_quat_plus.argtypes = [dp, dp, dp]
        L.orc_lm_solve.restype = C.c_int
        L.orc_lm_solve.argtypes = [dp, C.c_int, dp, dp, C.c_double, C.c_double, C.c_int, C.POINTER(LmTrace)]
        L.orc_cost.restype = C.c_double
        L.orc_cost.argtypes = [dp, C.c_int, dp, dp, C.c_double, C.c_double]
        L.orc_voxel_grid.restype = C.c_int64
        L.orc_voxel_grid.argtypes = [fp, C.c_int64, C.c_float, fp, C.c_int64]
        L.orc_transform.argtypes = [dp, fp, C.c_int64, fp]
        L.orc_pose_ops.argtypes = [dp, dp, dp, dp]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def make_params(min_range=3.0, max_range=75.0, lidar_type=0, scan_lines=64, scan_regions=8,
                edges_per_region=10, prev_frames=5, filter_local_map=False, mapping=False,
                lm_apply_step_on_ftol=0, knn_mode=0, pose_rotation_mode=1):
    """Defaults follow src/params.cc:40-109.  pose_rotation_mode: what Eigen's Transform::rotation()
    returns (laser_odometry.cc:164,186,403,420): 1 = Eigen 3.3.x polar factor (default: the README's
    platform ships Eigen 3.3.7), 0 = Eigen >= 3.4 alias of linear()."""
    p = OrcParams()
    p.min_range, p.max_range = min_range, max_range
    p.lidar_type, p.scan_lines = lidar_type, scan_lines
    p.scan_regions, p.edges_per_region = scan_regions, edges_per_region
    p.min_points_per_scan = scan_regions * edges_per_region + 10  # params.cc:63
    p.local_map_size = prev_frames
    # mapping: True / 1 = with the synchronous replay of the mapping node, 2 = ~map only via set_received_map
    p.filter_local_map, p.mapping = int(filter_local_map), int(mapping)
    p.lm_apply_step_on_ftol, p.knn_mode = lm_apply_step_on_ftol, knn_mode
    p.pose_rotation_mode = int(pose_rotation_mode)
    return p


def split(p, xyzi, height, width):
    xyzi = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
    n = xyzi.shape[0]
    offs = np.zeros(p.scan_lines + 1, dtype=np.int32)
    order = np.zeros(max(n, 1), dtype=np.int32)
    nv = lib().orc_split(C.byref(p), _fp(xyzi), n, height, width, _ip(offs), _ip(order))
    return offs, order[:nv]


def extract(p, xyzi, height, width, want_curv=False):
    """Returns dict(edges[E,4] float32, ring[E], idx_in_ring[E], src[E], curv[n_valid] or None)."""
    xyzi = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
    n = xyzi.shape[0]
    cap = p.scan_lines * p.scan_regions * (p.edges_per_region + 1) + 16
    edges = np.zeros((cap, 4), dtype=np.float32)
    ring = np.zeros(cap, dtype=np.int32)
    idx = np.zeros(cap, dtype=np.int32)
    src = np.zeros(cap, dtype=np.int32)
    curv = np.full(max(n, 1), np.nan, dtype=np.float64) if want_curv else None
    ne = lib().orc_extract(C.byref(p), _fp(xyzi), n, height, width, _fp(edges), _ip(ring), _ip(idx), _ip(src),
                           cap, _dp(curv) if want_curv else None)
    assert ne >= 0, "edge capacity too small"
    return dict(edges=edges[:ne].copy(), ring=ring[:ne].copy(), idx_in_ring=idx[:ne].copy(),
                src=src[:ne].copy(), curv=curv)


class Odometer:
    """LaserOdometer restatement (src/laser_odometry.cc:100-366)."""

    def __init__(self, p):
        self.p = p
        self.h = lib().orc_odom_create(C.byref(p))

    def close(self):
        if self.h:
            lib().orc_odom_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, edges):
        edges = np.ascontiguousarray(edges, dtype=np.float32).reshape(-1, 4)
        pose = np.zeros(7, dtype=np.float64)
        info = StepInfo()
        lib().orc_odom_step(self.h, _fp(edges), edges.shape[0], _dp(pose), C.byref(info))
        self.last_n = edges.shape[0]
        return pose, info

    def last_corr(self, it):
        n = self.last_n
        v = np.zeros(max(n, 1), dtype=np.int32)
        a = np.zeros(max(n, 1), dtype=np.int32)
        b = np.zeros(max(n, 1), dtype=np.int32)
        m = lib().orc_odom_last_corr(self.h, it, _ip(v), _ip(a), _ip(b), max(n, 1))
        m = max(m, 0)
        return v[:m], a[:m], b[:m]

    def last_queries(self, it):
        """World-frame float queries (edges x pose entering outer iteration `it`) of the last step."""
        n = self.last_n
        q = np.zeros((max(n, 1), 4), dtype=np.float32)
        m = max(lib().orc_odom_last_queries(self.h, it, _fp(q), max(n, 1)), 0)
        return q[:m, :3].copy()

    def window(self):
        n = lib().orc_odom_window_size(self.h)
        w = np.zeros((max(n, 1), 4), dtype=np.float32)
        lib().orc_odom_get_window(self.h, _fp(w), max(n, 1))
        return w[:n]

    def window_frames(self):
        return lib().orc_odom_window_frames(self.h)

    def received_map(self):
        cap = 1 << 20
        w = np.zeros((cap, 4), dtype=np.float32)
        n = lib().orc_odom_get_received_map(self.h, _fp(w), cap)
        return w[:max(n, 0)].copy()

    def map_total(self):
        return lib().orc_odom_map_total(self.h)

    def set_imu(self, q_xyzw, use_imu=True):
        """imuClb -> SharedData::setLastIMUOri (liodom_node.cc:66-70); use_imu = params->use_imu_."""
        q = np.ascontiguousarray(q_xyzw, dtype=np.float64)
        lib().orc_odom_set_imu(self.h, int(use_imu), _dp(q))

    def set_laser_to_base(self, T34):
        T = np.ascontiguousarray(T34, dtype=np.float64).reshape(12)
        lib().orc_odom_set_laser_to_base(self.h, _dp(T))

    def set_received_map(self, xyzi):
        xyzi = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
        lib().orc_odom_set_received_map(self.h, _fp(xyzi), xyzi.shape[0])

    def state(self):
        a = np.zeros(12)
        b = np.zeros(12)
        lib().orc_odom_get_state(self.h, _dp(a), _dp(b))
        return a.reshape(3, 4), b.reshape(3, 4)


class Map:
    """liodom::Map restatement (src/map.cc:70-189)."""

    def __init__(self, xy=40.0, z=50.0, res=0.4):
        self.h = lib().orc_map_create(xy, z, res)

    def __del__(self):
        try:
            lib().orc_map_destroy(self.h)
        except Exception:
            pass

    def update(self, xyzi, T34=None):
        x = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
        T = np.ascontiguousarray(np.eye(4)[:3] if T34 is None else T34, dtype=np.float64).reshape(12)
        lib().orc_map_update(self.h, _fp(x), x.shape[0], _dp(T))

    def local(self, T34=None, cells_xy=2, cells_z=1):
        T = np.ascontiguousarray(np.eye(4)[:3] if T34 is None else T34, dtype=np.float64).reshape(12)
        out = np.zeros((1 << 18, 4), dtype=np.float32)
        n = lib().orc_map_get_local(self.h, _dp(T), cells_xy, cells_z, _fp(out), out.shape[0])
        return out[:n].copy()

    def all(self):
        out = np.zeros((1 << 18, 4), dtype=np.float32)
        n = lib().orc_map_get_all(self.h, _fp(out), out.shape[0])
        return out[:n].copy()

    def num_cells(self):
        return lib().orc_map_num_cells(self.h)


def imu_override(T34, imu_q, l2b34=None, rotation_mode=1):
    """laser_odometry.cc:152-183 on one pose."""
    T = np.ascontiguousarray(T34, dtype=np.float64).reshape(12)
    q = np.ascontiguousarray(imu_q, dtype=np.float64)
    L = np.ascontiguousarray(np.eye(4)[:3] if l2b34 is None else l2b34, dtype=np.float64).reshape(12)
    out = np.zeros(12)
    lib().orc_imu_override(_dp(T), _dp(q), _dp(L), _dp(out), int(rotation_mode))
    return out.reshape(3, 4)


def publish_odom(prev34, cur34, dt, l2b34=None, rotation_mode=1):
    """publishOdom (laser_odometry.cc:395-436): orientation[4], position[3], linear[3], angular[3]."""
    P = np.ascontiguousarray(prev34, dtype=np.float64).reshape(12)
    T = np.ascontiguousarray(cur34, dtype=np.float64).reshape(12)
    L = np.ascontiguousarray(np.eye(4)[:3] if l2b34 is None else l2b34, dtype=np.float64).reshape(12)
    out = np.zeros(13)
    lib().orc_publish_odom(_dp(P), _dp(T), _dp(L), float(dt), _dp(out), int(rotation_mode))
    return out


def match_edges(p, map_xyzi, queries_xyz):
    """addEdgeConstraints' per-edge loop (laser_odometry.cc:320-361) on explicit inputs: (valid, ia, ib)."""
    m = np.ascontiguousarray(map_xyzi, dtype=np.float32).reshape(-1, 4)
    q3 = np.ascontiguousarray(queries_xyz, dtype=np.float32).reshape(-1, 3)
    q = np.zeros((q3.shape[0], 4), dtype=np.float32)
    q[:, :3] = q3
    n = q.shape[0]
    v, a, b = (np.zeros(max(n, 1), dtype=np.int32) for _ in range(3))
    lib().orc_match_edges(C.byref(p), _fp(m), m.shape[0], _fp(q), n, _ip(v), _ip(a), _ip(b))
    return v[:n], a[:n], b[:n]


def set_threads(stencil_threads=1, eval_threads=1):
    """Thread policy of the timed CPU baseline (results are identical for any setting).  The reference's:
    stencil = max(2, omp_get_max_threads() - 5) (feature_extractor.cc:29-34), eval = nproc (laser_odometry.cc:216)."""
    lib().orc_set_threads(int(stencil_threads), int(eval_threads))


def rotation_of(T34, mode=1):
    """Eigen::Transform::rotation() of a 3 x 4 pose (mode 1: Eigen 3.3 polar factor, 0: linear())."""
    T = np.ascontiguousarray(T34, dtype=np.float64).reshape(12)
    out = np.zeros(12)
    lib().orc_rotation_of(_dp(T), int(mode), _dp(out))
    return out.reshape(3, 4)


def knn5(map_xyzi, q_xyzi, mode=0):
    m = np.ascontiguousarray(map_xyzi, dtype=np.float32).reshape(-1, 4)
    q = np.ascontiguousarray(q_xyzi, dtype=np.float32).reshape(-1, 4)
    idx = np.zeros((q.shape[0], 5), dtype=np.int32)
    d = np.zeros((q.shape[0], 5), dtype=np.float32)
    lib().orc_knn5(_fp(m), m.shape[0], _fp(q), q.shape[0], mode, _ip(idx), _fp(d))
    return idx, d


def eig3(a6):
    a6 = np.ascontiguousarray(a6, dtype=np.float64)
    ev = np.zeros(3)
    lib().orc_eig3(_dp(a6), _dp(ev))
    return ev


def point2line(q, t, p, a, b, min_d=3.0, max_d=75.0):
    q, t, p, a, b = [np.ascontiguousarray(v, dtype=np.float64) for v in (q, t, p, a, b)]
    r = np.zeros(3)
    J = np.zeros(18)
    Jg = np.zeros(21)
    lib().orc_point2line(_dp(q), _dp(t), _dp(p), _dp(a), _dp(b), min_d, max_d, _dp(r), _dp(J), _dp(Jg))
    return r, J.reshape(3, 6), Jg.reshape(3, 7)


def quat_plus(x, delta):
    x = np.ascontiguousarray(x, dtype=np.float64)
    delta = np.ascontiguousarray(delta, dtype=np.float64)
    out = np.zeros(4)
    lib().orc_quat_plus(_dp(x), _dp(delta), _dp(out))
    return out


def lm_solve(blocks9, q, t, min_d=3.0, max_d=75.0, apply_on_ftol=0):
    b = np.ascontiguousarray(blocks9, dtype=np.float64).reshape(-1, 9)
    q = np.array(q, dtype=np.float64)
    t = np.array(t, dtype=np.float64)
    tr = LmTrace()
    lib().orc_lm_solve(_dp(b), b.shape[0], _dp(q), _dp(t), min_d, max_d, apply_on_ftol, C.byref(tr))
    return q, t, tr


def cost(blocks9, q, t, min_d=3.0, max_d=75.0):
    b = np.ascontiguousarray(blocks9, dtype=np.float64).reshape(-1, 9)
    q = np.ascontiguousarray(q, dtype=np.float64)
    t = np.ascontiguousarray(t, dtype=np.float64)
    return lib().orc_cost(_dp(b), b.shape[0], _dp(q), _dp(t), min_d, max_d)


def voxel_grid(xyzi, leaf):
    x = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
    out = np.zeros((max(x.shape[0], 1), 4), dtype=np.float32)
    n = lib().orc_voxel_grid(_fp(x), x.shape[0], leaf, _fp(out), max(x.shape[0], 1))
    return out[:n].copy()


def transform(T34, xyzi):
    T = np.ascontiguousarray(T34, dtype=np.float64).reshape(12)
    x = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
    out = np.zeros_like(x)
    lib().orc_transform(_dp(T), _fp(x), x.shape[0], _fp(out))
    return out


def pose_ops(q, t):
    q = np.ascontiguousarray(q, dtype=np.float64)
    t = np.ascontiguousarray(t, dtype=np.float64)
    T = np.zeros(12)
    qb = np.zeros(4)
    lib().orc_pose_ops(_dp(q), _dp(t), _dp(T), _dp(qb))
    return T.reshape(3, 4), qb

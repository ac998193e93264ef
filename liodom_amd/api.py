import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_LIB = os.path.join(_HERE, "lib", "libliodom_hip.so")
_HASH = _LIB + ".srchash"
_BUILD_INFO = {"rebuilt": None, "source_hash": None}
_SRC = [os.path.join(_HERE, "csrc", f) for f in ("liodom_hip.hip", "liodom_kernels.h", "kernels_extract.h", "kernels_sync.h",
                                                  "kernels_compact.h", "kernels_knn.h", "kernels_knn8.h", "kernels_lm.h", "kernels_rebuild.h",
                                                  "kernels_filter.h", "liodom_math.h", "wave_ops.h",
                                                  "liodom_map.h", "liodom_map_host.h")] + [
    os.path.join(_ROOT, "include", "liodom_hip.h")]

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wno-unused-value"]


class LiodomError(RuntimeError):
    pass


def lib_path():
    return _LIB


def source_hash():
    """SHA-256 over the compile flags and every source the library is built from."""
    h = hashlib.sha256()
    h.update(" ".join(HIPCC_FLAGS).encode())
    for p in _SRC:
        with open(p, "rb") as f:
            h.update(b"\0" + os.path.basename(p).encode() + b"\0" + f.read())
    return h.hexdigest()


def built_hash():
    """Source hash recorded beside the library when it was built ('' if none)."""
    try:
        with open(_HASH) as f:
            return f.read().strip()
    except OSError:
        return ""


def is_stale():
    return not os.path.exists(_LIB) or built_hash() != source_hash()


def build(force=False, verbose=False):
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU).  The library is
    rebuilt whenever the hash of its sources + flags differs from the one recorded at its last build
    (mtimes do not survive a copy to another box), so a stale binary is never tested against newer
    sources.  Returns the library path; build_info() tells whether this call compiled anything."""
    import fcntl
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    with open(_LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)           # several ranks may arrive here at once
        want = source_hash()
        if not force and os.path.exists(_LIB) and built_hash() == want:
            _BUILD_INFO.update(rebuilt=False, source_hash=want)
            return _LIB
        hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
        tmp = _LIB + ".tmp.%d" % os.getpid()
        cmd = [hipcc] + HIPCC_FLAGS + ["-o", tmp, _SRC[0]]
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, _LIB)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
        with open(_HASH, "w") as f:
            f.write(want + "\n")
        _BUILD_INFO.update(rebuilt=True, source_hash=want)
    return _LIB


def build_info():
    return dict(_BUILD_INFO)


class Params(C.Structure):
    """liodom_params_t — mirror of liodom::Params (include/liodom/params.h:33-49)."""
    _fields_ = [
        ("min_range", C.c_double), ("max_range", C.c_double),
        ("lidar_type", C.c_int32), ("scan_lines", C.c_int32), ("scan_regions", C.c_int32),
        ("edges_per_region", C.c_int32),
        ("min_points_per_scan", C.c_uint64), ("local_map_size", C.c_uint64),
        ("save_results", C.c_int32),
        ("results_dir", C.c_char * 256), ("fixed_frame", C.c_char * 64), ("base_frame", C.c_char * 64),
        ("laser_frame", C.c_char * 64),
        ("use_imu", C.c_int32), ("filter_local_map", C.c_int32), ("mapping", C.c_int32), ("publish_tf", C.c_int32),
    ]


class Config(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("n_streams", C.c_int32), ("max_points", C.c_int32), ("max_width", C.c_int32),
        ("reserved1", C.c_int32), ("lm_apply_step_on_ftol", C.c_int32), ("pose_log_capacity", C.c_int32),
        ("debug_buffers", C.c_int32), ("lm_workgroups", C.c_int32), ("recv_capacity", C.c_int32),
        ("pose_rotation_mode", C.c_int32), ("reserved0", C.c_int32),
    ]


class LmTrace(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("accepted", C.c_int32), ("termination", C.c_int32), ("pad", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double)]


class StepInfo(C.Structure):
    _fields_ = [("n_edges", C.c_int32), ("map_points", C.c_int32), ("matches", C.c_int32 * 2), ("lm", LmTrace * 2),
                ("status", C.c_uint32), ("scan_index", C.c_int32)]


NUM_KERNELS = 12


class MapConfig(C.Structure):
    """liodom_map_config_t — the mapping node's parameters (liodom_mapping_node.cc:115-125) + capacities."""
    _fields_ = [("device", C.c_int32), ("max_cells", C.c_int32),
                ("voxel_xysize", C.c_double), ("voxel_zsize", C.c_double), ("resolution", C.c_double),
                ("cell_capacity", C.c_int32), ("max_update_points", C.c_int32), ("max_modified_cells", C.c_int32),
                ("reserved", C.c_int32)]


class EdgeTicket(C.Structure):
    """liodom_edge_ticket_t: an edge cloud left on the device by liodom_extract_edges_device."""
    _fields_ = [("seq", C.c_uint32), ("slot", C.c_int32), ("stream", C.c_int32), ("reserved", C.c_int32)]


ERR_BUSY = -6
ERR_NEEDS_SYNC = -7


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_int64), ("total_ms", C.c_double)]


_lib = None


def load():
    """Load libliodom_hip.so.  Raises LiodomError if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if is_stale():
        # missing, or built from other sources than the ones present: rebuild (hipcc is part of the
        # image on the build box and on the GPU box); never load a binary that does not match the tree
        try:
            build()
        except Exception as ex:
            raise LiodomError("%s is missing or stale and could not be rebuilt (%s): run `python -c 'import "
                              "__graft_entry__ as g; g.build()'` (the HIP extension is required; there is no "
                              "CPU fallback)" % (_LIB, ex))
    L = C.CDLL(_LIB, mode=C.RTLD_GLOBAL)
    fp, dp, ip = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32)
    vp = C.c_void_p
    L.liodom_last_error.restype = C.c_char_p
    L.liodom_params_default.argtypes = [C.POINTER(Params)]
    L.liodom_config_default.argtypes = [C.POINTER(Config)]
    L.liodom_create.restype = C.c_int
    L.liodom_create.argtypes = [C.POINTER(Params), C.POINTER(Config), C.POINTER(vp)]
    L.liodom_destroy.argtypes = [vp]
    L.liodom_extract_edges.restype = C.c_int
    L.liodom_extract_edges.argtypes = [vp, C.c_int, fp, C.c_int64, C.c_int, C.c_int, fp, ip, ip, ip, C.c_int, ip]
    L.liodom_odometry_step.restype = C.c_int
    L.liodom_odometry_step.argtypes = [vp, C.c_int, fp, C.c_int, C.c_double, dp, C.POINTER(StepInfo)]
    L.liodom_process_scan.restype = C.c_int
    L.liodom_process_scan.argtypes = [vp, C.c_int, fp, C.c_int64, C.c_int, C.c_int, C.c_double, dp, C.POINTER(StepInfo)]
    L.liodom_set_received_map.restype = C.c_int
    L.liodom_set_received_map.argtypes = [vp, C.c_int, fp, C.c_int64]
    L.liodom_alloc_resident.restype = C.c_int
    L.liodom_alloc_resident.argtypes = [vp, C.c_int]
    L.liodom_upload_scan.restype = C.c_int
    L.liodom_upload_scan.argtypes = [vp, C.c_int, C.c_int, fp, C.c_int64]
    L.liodom_process_resident.restype = C.c_int
    L.liodom_process_resident.argtypes = [vp, C.c_int, C.c_int64, C.c_int, C.c_int, dp, C.POINTER(StepInfo)]
    L.liodom_process_resident_pipelined.restype = C.c_int
    L.liodom_replay_resident.restype = C.c_int
    L.liodom_replay_resident.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, dp, C.POINTER(StepInfo)]
    L.liodom_process_resident_pipelined.argtypes = [vp, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, dp, C.POINTER(StepInfo)]
    L.liodom_sync.restype = C.c_int
    L.liodom_sync.argtypes = [vp]
    L.liodom_get_pose_log.restype = C.c_int
    L.liodom_get_pose_log.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp, C.POINTER(StepInfo)]
    L.liodom_reset.restype = C.c_int
    L.liodom_reset.argtypes = [vp]
    L.liodom_get_edges.restype = C.c_int
    L.liodom_get_edges.argtypes = [vp, C.c_int, fp, ip, ip, ip, C.c_int, ip]
    L.liodom_get_window.restype = C.c_int
    L.liodom_get_window.argtypes = [vp, C.c_int, fp, C.c_int64, C.POINTER(C.c_int64), ip]
    L.liodom_get_local_map.restype = C.c_int
    L.liodom_get_local_map.argtypes = [vp, C.c_int, fp, C.c_int64, C.POINTER(C.c_int64), ip]
    L.liodom_get_correspondences.restype = C.c_int
    L.liodom_get_correspondences.argtypes = [vp, C.c_int, C.c_int, ip, ip, ip, C.c_int, ip]
    L.liodom_get_knn_queries.restype = C.c_int
    L.liodom_get_knn_queries.argtypes = [vp, C.c_int, C.c_int, fp, C.c_int, ip]
    L.liodom_get_curvature.restype = C.c_int
    L.liodom_get_curvature.argtypes = [vp, C.c_int, dp, C.c_int64, ip]
    L.liodom_set_profiling.restype = C.c_int
    L.liodom_set_profiling.argtypes = [vp, C.c_int]
    L.liodom_get_kernel_stats.restype = C.c_int
    L.liodom_get_kernel_stats.argtypes = [vp, C.POINTER(KernelStat)]
    L.liodom_reset_kernel_stats.restype = C.c_int
    L.liodom_reset_kernel_stats.argtypes = [vp]
    L.liodom_device_count.restype = C.c_int
    L.liodom_device_count.argtypes = [ip]
    L.liodom_device_pci_bus_id.restype = C.c_int
    L.liodom_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_int]
    L.liodom_device_info.restype = C.c_int
    L.liodom_device_info.argtypes = [vp, C.c_char_p, C.c_int, ip]
    i64p = C.POINTER(C.c_int64)
    L.liodom_get_received_map.restype = C.c_int
    L.liodom_get_received_map.argtypes = [vp, C.c_int, fp, C.c_int64, i64p]
    L.liodom_set_imu_orientation.restype = C.c_int
    L.liodom_set_imu_orientation.argtypes = [vp, C.c_int, dp]
    L.liodom_set_laser_to_base.restype = C.c_int
    L.liodom_set_laser_to_base.argtypes = [vp, dp]
    L.liodom_attach_mapper.restype = C.c_int
    L.liodom_attach_mapper.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int]
    L.liodom_map_config_default.argtypes = [C.POINTER(MapConfig)]
    L.liodom_map_create.restype = C.c_int
    L.liodom_map_create.argtypes = [C.POINTER(MapConfig), C.POINTER(vp)]
    L.liodom_map_destroy.argtypes = [vp]
    L.liodom_map_update.restype = C.c_int
    L.liodom_map_update.argtypes = [vp, fp, C.c_int64, dp]
    L.liodom_map_get_local.restype = C.c_int
    L.liodom_map_get_local.argtypes = [vp, dp, C.c_int, C.c_int, fp, C.c_int64, i64p]
    L.liodom_map_get_all.restype = C.c_int
    L.liodom_map_get_all.argtypes = [vp, fp, C.c_int64, i64p]
    L.liodom_map_num_cells.restype = C.c_int
    L.liodom_map_num_cells.argtypes = [vp, ip]
    L.liodom_map_status.restype = C.c_int
    L.liodom_map_status.argtypes = [vp, C.POINTER(C.c_uint32)]
    tp = C.POINTER(EdgeTicket)
    L.liodom_scan_buffer.restype = C.c_int
    L.liodom_scan_buffer.argtypes = [vp, C.c_int, C.POINTER(fp), i64p]
    L.liodom_extract_edges_device.restype = C.c_int
    L.liodom_extract_edges_device.argtypes = [vp, C.c_int, fp, C.c_int64, C.c_int, C.c_int, tp]
    L.liodom_wait_edges.restype = C.c_int
    L.liodom_wait_edges.argtypes = [vp, tp, fp, ip, ip, ip, C.c_int, ip]
    L.liodom_odometry_step_device.restype = C.c_int
    L.liodom_odometry_step_device.argtypes = [vp, tp, C.c_double, dp, C.POINTER(StepInfo)]
    L.liodom_odometry_submit_device.restype = C.c_int
    L.liodom_odometry_submit_device.argtypes = [vp, tp, C.c_double]
    L.liodom_odometry_collect.restype = C.c_int
    L.liodom_odometry_collect.argtypes = [vp, C.c_int, dp, C.POINTER(StepInfo)]
    L.liodom_pin_host_buffer.restype = C.c_int
    L.liodom_pin_host_buffer.argtypes = [C.c_void_p, C.c_int64]
    L.liodom_unpin_host_buffer.restype = C.c_int
    L.liodom_unpin_host_buffer.argtypes = [C.c_void_p]
    _lib = L
    return L


_host_lib = None


def host_lib():
    """libliodom_host.so: the C++ mirror of the reference's classes over the C-ABI (liodom_amd/host), built by make."""
    global _host_lib
    if _host_lib is None:
        load()
        hd = os.path.join(_HERE, "host")
        subprocess.check_call(["make", "-C", hd, "-s"])
        HL = C.CDLL(os.path.join(hd, "libliodom_host.so"), mode=C.RTLD_GLOBAL)
        HL.liodom_host_two_thread_replay.restype = C.c_int
        HL.liodom_host_two_thread_replay.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int,
                                                     C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                     C.POINTER(C.c_int64)]
        _host_lib = HL
    return _host_lib


EXPORTED_SYMBOLS = [
    "liodom_params_default", "liodom_config_default", "liodom_create", "liodom_destroy", "liodom_last_error",
    "liodom_extract_edges", "liodom_odometry_step", "liodom_process_scan", "liodom_set_received_map",
    "liodom_alloc_resident", "liodom_upload_scan", "liodom_process_resident", "liodom_process_resident_pipelined", "liodom_replay_resident", "liodom_sync", "liodom_get_pose_log",
    "liodom_reset", "liodom_get_edges", "liodom_get_window", "liodom_get_local_map", "liodom_get_correspondences", "liodom_get_curvature", "liodom_get_knn_queries",
    "liodom_set_profiling", "liodom_get_kernel_stats", "liodom_reset_kernel_stats", "liodom_device_info",
    "liodom_device_count", "liodom_device_pci_bus_id", "liodom_get_modes", "liodom_replay_host", "liodom_pin_host_buffer", "liodom_unpin_host_buffer",
    "liodom_map_config_default", "liodom_map_create", "liodom_map_destroy", "liodom_map_update", "liodom_map_get_local",
    "liodom_map_get_all", "liodom_map_num_cells", "liodom_map_status", "liodom_get_received_map", "liodom_attach_mapper",
    "liodom_set_imu_orientation", "liodom_set_laser_to_base",
    "liodom_scan_buffer", "liodom_extract_edges_device", "liodom_wait_edges", "liodom_odometry_step_device",
    "liodom_odometry_submit_device", "liodom_odometry_collect",
]


def device_count():
    n = C.c_int32()
    load().liodom_device_count(C.byref(n))
    return n.value


def device_pci_bus_id(device):
    buf = C.create_string_buffer(64)
    if load().liodom_device_pci_bus_id(int(device), buf, 64) != 0:
        return None
    return buf.value.decode()


def make_params(**kw):
    """Params with the defaults of Params::readParams (src/params.cc:40-109); `prev_frames`
    is the ROS parameter behind local_map_size."""
    p = Params()
    load().liodom_params_default(C.byref(p))
    if "prev_frames" in kw:
        kw["local_map_size"] = kw.pop("prev_frames")
    for k, v in kw.items():
        setattr(p, k, v)
    if "min_points_per_scan" not in kw:
        p.min_points_per_scan = p.scan_regions * p.edges_per_region + 10  # params.cc:63
    return p


def make_config(**kw):
    c = Config()
    load().liodom_config_default(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Liodom:
    """One handle = one GPU, `n_streams` lock-step LiDAR streams."""

    def __init__(self, params, config):
        self.L = load()
        self.params, self.config = params, config
        h = C.c_void_p()
        rc = self.L.liodom_create(C.byref(params), C.byref(config), C.byref(h))
        if rc != 0:
            raise LiodomError("liodom_create failed (%d): %s" % (rc, self.L.liodom_last_error().decode()))
        self.h = h
        self.edge_cap = params.scan_lines * params.scan_regions * (params.edges_per_region + 1) + 64

    def _check(self, rc):
        if rc != 0:
            raise LiodomError("libliodom_hip error %d: %s" % (rc, self.L.liodom_last_error().decode()))

    def close(self):
        if getattr(self, "h", None):
            self.L.liodom_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- FeatureExtractor ---
    def extract_edges(self, xyzi, height, width, stream=0):
        x = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
        cap = self.edge_cap
        e = np.zeros((cap, 4), np.float32)
        ring, idx, src = (np.zeros(cap, np.int32) for _ in range(3))
        n = C.c_int32()
        self._check(self.L.liodom_extract_edges(self.h, stream, _fp(x), x.shape[0], height, width, _fp(e), _ip(ring),
                                                _ip(idx), _ip(src), cap, C.byref(n)))
        k = n.value
        return dict(edges=e[:k].copy(), ring=ring[:k].copy(), idx_in_ring=idx[:k].copy(), src=src[:k].copy())

    def get_edges(self, stream=0):
        cap = self.edge_cap
        e = np.zeros((cap, 4), np.float32)
        ring, idx, src = (np.zeros(cap, np.int32) for _ in range(3))
        n = C.c_int32()
        self._check(self.L.liodom_get_edges(self.h, stream, _fp(e), _ip(ring), _ip(idx), _ip(src), cap, C.byref(n)))
        k = n.value
        return dict(edges=e[:k].copy(), ring=ring[:k].copy(), idx_in_ring=idx[:k].copy(), src=src[:k].copy())

    # --- LaserOdometer ---
    def odometry_step(self, edges, stamp=0.0, stream=0):
        e = np.ascontiguousarray(edges, dtype=np.float32).reshape(-1, 4)
        pose = np.zeros(7)
        info = StepInfo()
        self._check(self.L.liodom_odometry_step(self.h, stream, _fp(e), e.shape[0], stamp, _dp(pose), C.byref(info)))
        return pose, info

    # --- the same two sides with the edge cloud staying on the device (tickets) ---
    def scan_buffer(self, stream=0):
        """Page-locked buffer [max_points, 4] to assemble the next cloud in (liodom_scan_buffer)."""
        p = C.POINTER(C.c_float)()
        cap = C.c_int64()
        self._check(self.L.liodom_scan_buffer(self.h, stream, C.byref(p), C.byref(cap)))
        return np.ctypeslib.as_array(p, shape=(cap.value, 4))

    def extract_edges_device(self, xyzi, height, width, stream=0):
        """Enqueues upload + extraction; returns the ticket, or None when every hand-off slot is taken (LIODOM_ERR_BUSY).
        A page-locked `xyzi` (scan_buffer / pin) must stay untouched until wait_edges or the ticket's odometry returned."""
        x = xyzi if (isinstance(xyzi, np.ndarray) and xyzi.dtype == np.float32 and xyzi.flags["C_CONTIGUOUS"]) else np.ascontiguousarray(xyzi, dtype=np.float32)
        n = x.size // 4
        t = EdgeTicket()
        rc = self.L.liodom_extract_edges_device(self.h, stream, _fp(x), n, height, width, C.byref(t))
        if rc == ERR_BUSY:
            return None
        self._check(rc)
        t._keep = x          # the upload may still be reading it
        return t

    def wait_edges(self, ticket):
        cap = self.edge_cap
        e = np.zeros((cap, 4), np.float32)
        ring, idx, src = (np.zeros(cap, np.int32) for _ in range(3))
        n = C.c_int32()
        self._check(self.L.liodom_wait_edges(self.h, C.byref(ticket), _fp(e), _ip(ring), _ip(idx), _ip(src), cap, C.byref(n)))
        k = n.value
        return dict(edges=e[:k].copy(), ring=ring[:k].copy(), idx_in_ring=idx[:k].copy(), src=src[:k].copy())

    def odometry_step_device(self, ticket, stamp=0.0):
        pose = np.zeros(7)
        info = StepInfo()
        self._check(self.L.liodom_odometry_step_device(self.h, C.byref(ticket), stamp, _dp(pose), C.byref(info)))
        return pose, info

    def odometry_submit_device(self, ticket, stamp=0.0):
        """False when two scans are already in flight (LIODOM_ERR_BUSY)."""
        rc = self.L.liodom_odometry_submit_device(self.h, C.byref(ticket), stamp)
        if rc == ERR_BUSY:
            return False
        self._check(rc)
        return True

    def odometry_collect(self, stream=0):
        pose = np.zeros(7)
        info = StepInfo()
        self._check(self.L.liodom_odometry_collect(self.h, stream, _dp(pose), C.byref(info)))
        return pose, info

    def two_thread_replay(self, scans, n, height, width, timed_from=0, fetch_edges=True, depth=1, pin=True):
        """The two-thread binding driven by two C++ threads (liodom_host_two_thread_replay in libliodom_host.so: an extractor
        thread with liodom_extract_edges_device + liodom_wait_edges, an odometer thread with liodom_odometry_submit_device /
        liodom_odometry_collect, a ticket queue between them).  scans: float32 [count, max_points, 4] in host memory (page-locked
        for the call if pin).  Returns (poses [count, 7], seconds from the submission of scan timed_from to the last pose,
        total number of edges fetched)."""
        HL = host_lib()
        a = np.ascontiguousarray(scans, dtype=np.float32)
        count = a.shape[0]
        stride = int(a.size // max(1, count))
        poses = np.zeros((count, 7))
        secs = C.c_double()
        tot = C.c_int64()
        pinned = pin and self.L.liodom_pin_host_buffer(a.ctypes.data_as(C.c_void_p), a.nbytes) == 0
        try:
            self._check(HL.liodom_host_two_thread_replay(self.h, _fp(a), stride, count, n, height, width, int(timed_from),
                                                         1 if fetch_edges else 0, int(depth), self.edge_cap, _dp(poses),
                                                         C.byref(secs), C.byref(tot)))
        finally:
            if pinned:
                self.L.liodom_unpin_host_buffer(a.ctypes.data_as(C.c_void_p))
        return poses, secs.value, tot.value

    def process_scan(self, xyzi, height, width, stamp=0.0, stream=0):
        x = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
        pose = np.zeros(7)
        info = StepInfo()
        self._check(self.L.liodom_process_scan(self.h, stream, _fp(x), x.shape[0], height, width, stamp, _dp(pose),
                                               C.byref(info)))
        return pose, info

    # --- resident replay ---
    def alloc_resident(self, n_slots):
        self._check(self.L.liodom_alloc_resident(self.h, n_slots))

    def upload_scan(self, stream, slot, xyzi):
        x = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
        self._check(self.L.liodom_upload_scan(self.h, stream, slot, _fp(x), x.shape[0]))

    def process_resident(self, slot, n, height, width, readback=True, next_slot=-1):
        """next_slot >= 0: also issue that slot's extraction on the second stream (pipelined replay)."""
        S = self.config.n_streams
        if readback:
            poses = np.zeros((S, 7))
            infos = (StepInfo * S)()
            self._check(self.L.liodom_process_resident_pipelined(self.h, slot, next_slot, n, height, width, _dp(poses), infos))
            return poses, infos
        self._check(self.L.liodom_process_resident_pipelined(self.h, slot, next_slot, n, height, width, None, None))
        return None, None

    def replay_resident(self, first_slot, count, n, height, width, ahead=False, depth=0):
        """The pipelined consumer loop over resident slots first_slot .. first_slot + count - 1, in C
        (liodom_replay_resident): every pose read back in order; depth 0 = strictly synchronous, 1 = the odometry of
        scan k+1 is submitted before pose k is waited for.  Returns poses [count, n_streams, 7] and the step infos."""
        S = self.config.n_streams
        poses = np.zeros((count, S, 7))
        infos = (StepInfo * (count * S))()
        self._check(self.L.liodom_replay_resident(self.h, first_slot, count, 1 if ahead else 0, int(depth), n, height, width, _dp(poses), infos))
        return poses, infos

    def replay_host(self, scans, n, height, width, depth=1, pin=True):
        """Host-fed replay (liodom_replay_host): scans = float32 array [count, n_streams, max_points, 4] in host memory
        (page-locked for the duration of the call if pin).  Returns poses [count, n_streams, 7] and the step infos."""
        S = self.config.n_streams
        a = np.ascontiguousarray(scans, dtype=np.float32)
        count = a.shape[0]
        stride = int(a.size // max(1, count * S))
        poses = np.zeros((count, S, 7))
        infos = (StepInfo * (count * S))()
        self.L.liodom_replay_host.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(StepInfo)]
        self.L.liodom_pin_host_buffer.argtypes = [C.c_void_p, C.c_int64]
        self.L.liodom_unpin_host_buffer.argtypes = [C.c_void_p]
        pinned = pin and self.L.liodom_pin_host_buffer(a.ctypes.data_as(C.c_void_p), a.nbytes) == 0
        try:
            self._check(self.L.liodom_replay_host(self.h, _fp(a), stride, count, int(depth), n, height, width, _dp(poses), infos))
        finally:
            if pinned:
                self.L.liodom_unpin_host_buffer(a.ctypes.data_as(C.c_void_p))
        return poses, infos

    def modes(self):
        """The code paths this handle runs ("key=value ..." from liodom_get_modes) as a dict of strings."""
        buf = C.create_string_buffer(1024)
        self.L.liodom_get_modes.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        self._check(self.L.liodom_get_modes(self.h, buf, 1024))
        return dict(kv.split("=", 1) for kv in buf.value.decode().split())

    def sync(self):
        self._check(self.L.liodom_sync(self.h))

    def reset(self):
        self._check(self.L.liodom_reset(self.h))

    def pose_log(self, stream, first, count):
        poses = np.zeros((count, 7))
        infos = (StepInfo * count)()
        self._check(self.L.liodom_get_pose_log(self.h, stream, first, count, _dp(poses), infos))
        return poses, infos

    # --- inspection ---
    def window(self, stream=0):
        cap = self.edge_cap * int(self.params.local_map_size)
        w = np.zeros((cap, 4), np.float32)
        n = C.c_int64()
        nf = C.c_int32()
        self._check(self.L.liodom_get_window(self.h, stream, _fp(w), cap, C.byref(n), C.byref(nf)))
        return w[:n.value].copy(), nf.value

    def set_imu(self, q_xyzw, stream=0):
        """imuClb (liodom_node.cc:66-70): latest IMU orientation, used when params.use_imu."""
        q = np.ascontiguousarray(q_xyzw, dtype=np.float64)
        self._check(self.L.liodom_set_imu_orientation(self.h, stream, _dp(q)))

    def set_laser_to_base(self, T34):
        T = np.ascontiguousarray(T34, dtype=np.float64).reshape(12)
        self._check(self.L.liodom_set_laser_to_base(self.h, _dp(T)))

    def set_received_map(self, xyzi, stream=0):
        """mapClb (liodom_node.cc:57-64): the cloud the mapper published on ~map."""
        x = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
        self._check(self.L.liodom_set_received_map(self.h, stream, _fp(x), x.shape[0]))

    def received_map(self, stream=0):
        cap = max(int(self.config.recv_capacity), 262144)
        w = np.zeros((cap, 4), np.float32)
        n = C.c_int64()
        self._check(self.L.liodom_get_received_map(self.h, stream, _fp(w), cap, C.byref(n)))
        return w[:n.value].copy()

    def attach_mapper(self, mapper, cells_xy=2, cells_z=1, stream=0):
        """Synchronous on-device replay of the liodom_mapping node for this stream (see
        liodom_attach_mapper in include/liodom_hip.h)."""
        self._check(self.L.liodom_attach_mapper(self.h, stream, mapper.h if mapper is not None else None, cells_xy, cells_z))
        self._mappers = getattr(self, "_mappers", {})
        self._mappers[stream] = mapper      # keep it alive while attached

    def local_map(self, stream=0):
        cap = self.edge_cap * int(self.params.local_map_size) + (max(int(self.config.recv_capacity), 262144) if self.params.mapping else 0)
        w = np.zeros((cap, 4), np.float32)
        n = C.c_int64()
        filt = C.c_int32()
        self._check(self.L.liodom_get_local_map(self.h, stream, _fp(w), cap, C.byref(n), C.byref(filt)))
        return w[:n.value].copy(), bool(filt.value)

    def correspondences(self, it, stream=0):
        cap = self.edge_cap
        v, a, b = (np.zeros(cap, np.int32) for _ in range(3))
        n = C.c_int32()
        self._check(self.L.liodom_get_correspondences(self.h, stream, it, _ip(v), _ip(a), _ip(b), cap, C.byref(n)))
        k = n.value
        return v[:k].copy(), a[:k].copy(), b[:k].copy()

    def knn_queries(self, it, stream=0):
        """World-frame float queries of outer iteration `it` of the last step (debug_buffers = 1)."""
        cap = self.edge_cap
        q = np.zeros((cap, 4), np.float32)
        n = C.c_int32()
        self._check(self.L.liodom_get_knn_queries(self.h, stream, it, _fp(q), cap, C.byref(n)))
        return q[:n.value, :3].copy()

    def curvature(self, stream=0):
        cap = int(self.config.max_points) + 16
        c = np.zeros(cap)
        offs = np.zeros(self.params.scan_lines + 1, np.int32)
        self._check(self.L.liodom_get_curvature(self.h, stream, _dp(c), cap, _ip(offs)))
        return c[:offs[-1]].copy(), offs

    # --- measurement ---
    def set_profiling(self, on):
        self._check(self.L.liodom_set_profiling(self.h, int(on)))

    def kernel_stats(self):
        st = (KernelStat * NUM_KERNELS)()
        self._check(self.L.liodom_get_kernel_stats(self.h, st))
        return {s.name.decode(): (int(s.launches), float(s.total_ms)) for s in st}

    def reset_kernel_stats(self):
        self._check(self.L.liodom_reset_kernel_stats(self.h))

    def device_info(self):
        buf = C.create_string_buffer(256)
        cu = C.c_int32()
        self._check(self.L.liodom_device_info(self.h, buf, 256, C.byref(cu)))
        return buf.value.decode(), cu.value


class Map:
    """liodom::Map on the device (src/map.cc): updateMap / getLocalMap / getMap of the mapping node."""

    def __init__(self, xy=40.0, z=50.0, res=0.4, **caps):
        L = load()
        c = MapConfig()
        L.liodom_map_config_default(C.byref(c))
        c.voxel_xysize, c.voxel_zsize, c.resolution = xy, z, res
        for k, v in caps.items():
            if not hasattr(c, k):
                raise KeyError(k)
            setattr(c, k, v)
        self.h = C.c_void_p()
        self._L = L
        self._chk(L.liodom_map_create(C.byref(c), C.byref(self.h)))
        self.result_capacity = 1 << 18

    def _chk(self, rc):
        if rc != 0:
            raise LiodomError("liodom_map error %d: %s" % (rc, (self._L.liodom_last_error() or b"").decode()))

    def close(self):
        if self.h:
            self._L.liodom_map_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _T(T34):
        return np.ascontiguousarray(np.eye(4)[:3] if T34 is None else T34, dtype=np.float64).reshape(12)

    def update(self, xyzi, T34=None):
        x = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
        T = self._T(T34)
        self._chk(self._L.liodom_map_update(self.h, _fp(x), x.shape[0], _dp(T)))

    def local(self, T34=None, cells_xy=2, cells_z=1):
        T = self._T(T34)
        out = np.zeros((self.result_capacity, 4), dtype=np.float32)
        n = C.c_int64(0)
        self._chk(self._L.liodom_map_get_local(self.h, _dp(T), cells_xy, cells_z, _fp(out), out.shape[0], C.byref(n)))
        return out[:n.value].copy()

    def all(self):
        out = np.zeros((self.result_capacity, 4), dtype=np.float32)
        n = C.c_int64(0)
        self._chk(self._L.liodom_map_get_all(self.h, _fp(out), out.shape[0], C.byref(n)))
        return out[:n.value].copy()

    def num_cells(self):
        n = C.c_int32(0)
        self._chk(self._L.liodom_map_num_cells(self.h, C.byref(n)))
        return n.value

    def status(self):
        s = C.c_uint32(0)
        self._chk(self._L.liodom_map_status(self.h, C.byref(s)))
        return s.value

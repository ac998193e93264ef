"""Synthetic LiDAR stream generator (liodom_amd/synth/synth.cc) — bench / test input.

The stream definition follows SURVEY.md §8(d); see the header of synth.cc.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libliodom_synth.so")


class SynthCfg(C.Structure):
    _fields_ = [
        ("height", C.c_int32), ("width", C.c_int32), ("lidar_type", C.c_int32), ("world_seed", C.c_uint32),
        ("noise_sigma", C.c_double), ("max_cast_range", C.c_double),
        ("yaw_rate_deg", C.c_double), ("speed", C.c_double),
    ]


def build(force=False):
    src = os.path.join(_HERE, "synth.cc")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O3", "-std=c++17", "-fPIC", "-fopenmp", "-shared", "-o", _LIB_PATH, src])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.synth_scan.restype = C.c_int
        L.synth_scan.argtypes = [C.POINTER(SynthCfg), C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]
        L.synth_num_boxes.restype = C.c_int
        L.synth_num_boxes.argtypes = []
        _lib = L
    return _lib


def make_cfg(height, width, lidar_type=0, world_seed=7, noise_sigma=0.01, max_cast_range=120.0,
             yaw_rate_deg=0.5, speed=0.1):
    c = SynthCfg()
    c.height, c.width, c.lidar_type, c.world_seed = height, width, lidar_type, world_seed
    c.noise_sigma, c.max_cast_range = noise_sigma, max_cast_range
    c.yaw_rate_deg, c.speed = yaw_rate_deg, speed
    return c


def scan(cfg, stream, k):
    """Returns (xyzi[N,4] float32, gt_pose[7] = qx qy qz qw tx ty tz)."""
    out = np.zeros((cfg.height * cfg.width, 4), dtype=np.float32)
    pose = np.zeros(7, dtype=np.float64)
    rc = lib().synth_scan(C.byref(cfg), stream, k, out.ctypes.data_as(C.POINTER(C.c_float)),
                          pose.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return out, pose


def ragged(xyzi, height, width, lidar_type=0, seed=0, min_keep=0.55, dead_rings=3):
    """A ragged copy of a synthetic scan (what real sensors deliver and SURVEY.md §8(d)'s box world never does): 20-30 % of the
    returns become NaN no-returns (isValidPoint, feature_extractor.cc:84-102), every ring loses its own share — between 5 % and
    1 - min_keep of its points, in bursts and singly — so that the compacted rings have unequal lengths, and `dead_rings` rings
    keep fewer than min_points_per_scan points (skipped at feature_extractor.cc:188-190).  The ring of a point is its beam:
    firing order is column-major for lidar_type 0 (point i = column i // height, beam i % height), row-major for lidar_type 1."""
    rng = np.random.default_rng(1234567 + int(seed))
    out = np.array(xyzi, dtype=np.float32, copy=True).reshape(-1, 4)
    n = out.shape[0]
    idx = np.arange(n)
    beam = (idx % height) if lidar_type == 0 else (idx // width)
    col = (idx // height) if lidar_type == 0 else (idx % width)
    drop = np.zeros(n, dtype=bool)
    frac = rng.uniform(0.05, 1.0 - min_keep, size=height)
    dead = rng.choice(height, size=min(dead_rings, height), replace=False)
    for b in range(height):
        m = beam == b
        if b in dead:
            keep_cols = rng.choice(width, size=int(rng.integers(5, 60)), replace=False)
            drop[m] = ~np.isin(col[m], keep_cols)
            continue
        d = rng.random(width) < 0.5 * frac[b]                    # single no-returns
        nb = max(1, int(0.5 * frac[b] * width / 40))             # bursts of ~40 columns (sky, glass)
        for s in rng.integers(0, width, size=nb):
            d[s:s + int(rng.integers(10, 70))] = True
        drop[m] = d[col[m]]
    out[drop, :3] = np.nan
    return out

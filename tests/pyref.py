"""Independent pure-Python/NumPy transcription of the edge-selection logic.

Written from the reference's control flow (sort, then walk) rather than from the oracle's or the
GPU's formulation, so that it cross-checks both.  Cites /root/reference paths.  Slow: use on
small clouds only.
"""
import math

import numpy as np


def is_valid(x, y, z, min_range, max_range):
    # src/feature_extractor.cc:84-102
    ok = math.isfinite(x) and math.isfinite(y) and math.isfinite(z)
    d = math.sqrt(x * x + y * y) if ok else float("nan")
    if not ok:
        return False, d
    if d > max_range or d < min_range:
        return False, d
    return True, d


def velodyne_ring(z, dist, scan_lines):
    # src/feature_extractor.cc:127-151
    angle = math.atan(z / dist) * 180 / math.pi
    if scan_lines == 64:
        if angle >= -8.83:
            sid = int((2 - angle) * 3.0 + 0.5)
        else:
            sid = scan_lines // 2 + int((-8.83 - angle) * 2.0 + 0.5)
        if angle > 2 or angle < -24.33 or sid > 63 or sid < 0:
            return -1
        return sid
    if scan_lines == 32:
        sid = int((angle + 92.0 / 3.0) * 3.0 / 4.0)
        return sid if 0 <= sid <= 31 else -1
    if scan_lines == 16:
        sid = int((angle + 15) / 2 + 0.5)
        return sid if 0 <= sid <= 15 else -1
    return -1


def split(xyzi, height, width, lidar_type, scan_lines, min_range, max_range):
    rings = [[] for _ in range(scan_lines)]
    if lidar_type == 0:
        for i in range(xyzi.shape[0]):
            x, y, z = float(xyzi[i, 0]), float(xyzi[i, 1]), float(xyzi[i, 2])
            ok, d = is_valid(x, y, z, min_range, max_range)
            if not ok:
                continue
            r = velodyne_ring(z, d, scan_lines)
            if r != -1:
                rings[r].append(i)
    else:
        for row in range(height):
            for col in range(width):
                i = row * width + col
                ok, _ = is_valid(float(xyzi[i, 0]), float(xyzi[i, 1]), float(xyzi[i, 2]), min_range, max_range)
                if ok and row < scan_lines:
                    rings[row].append(i)
    return rings


def extract(xyzi, height, width, lidar_type=0, scan_lines=64, scan_regions=8, edges_per_region=10,
            min_range=3.0, max_range=75.0):
    """Returns list of (ring, idx_in_ring, src_index) in the reference's output order."""
    min_points = scan_regions * edges_per_region + 10          # src/params.cc:63
    rings = split(xyzi, height, width, lidar_type, scan_lines, min_range, max_range)
    out = []
    for r in range(scan_lines):
        src = rings[r]
        n = len(src)
        if n < min_points:                                     # feature_extractor.cc:188
            continue
        # The operands are pcl::PointXYZI floats and `10 * x` is int * float: the eleven-term sum
        # is evaluated in float32, left to right, and only then widened to double (:196-228).
        P = xyzi[src, :3].astype(np.float32)
        ten = np.float32(10)
        smooth = {}
        picked = [False] * n
        for j in range(5, n - 5):                              # :195-232, left-to-right sums
            d = [0.0, 0.0, 0.0]
            for ax in range(3):
                s = P[j - 5, ax]                               # np.float32 scalars: every op rounds to float32
                s = s + P[j - 4, ax]
                s = s + P[j - 3, ax]
                s = s + P[j - 2, ax]
                s = s + P[j - 1, ax]
                s = s - ten * P[j, ax]
                s = s + P[j + 1, ax]
                s = s + P[j + 2, ax]
                s = s + P[j + 3, ax]
                s = s + P[j + 4, ax]
                s = s + P[j + 5, ax]
                assert s.dtype == np.float32
                d[ax] = float(s)                               # double diff_x = <float expression>
            smooth[j] = float(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])   # :229 in double
        total = n - 10
        sector = total // scan_regions
        for reg in range(scan_regions):                        # :240-252
            start, end = sector * reg, sector * (reg + 1)
            if reg == scan_regions - 1:
                end = total
            items = [(k + 5, smooth[k + 5]) for k in range(start, end)]
            items.sort(key=lambda it: (-it[1], it[0]))         # std::sort desc, ties: index asc
            npick = 0
            for (pi, sm) in items:                             # :265-312
                if picked[pi]:
                    continue
                if sm < 0.1 or npick > edges_per_region:
                    break
                out.append((r, pi, src[pi]))
                npick += 1
                picked[pi] = True
                for l in range(1, 6):
                    dd = (P[pi + l] - P[pi + l - 1]).astype(np.float64)   # float32 difference, then double (:281-286)
                    if dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2] > 0.05:
                        break
                    picked[pi + l] = True
                for l in range(-1, -6, -1):
                    dd = (P[pi + l] - P[pi + l + 1]).astype(np.float64)   # :297-302
                    if dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2] > 0.05:
                        break
                    picked[pi + l] = True
    return out

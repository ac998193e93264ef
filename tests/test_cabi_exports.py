"""The C-ABI library builds for gfx950, loads without a GPU, exports every function declared in
include/liodom_hip.h, mirrors liodom::Params field for field, and refuses to run without a
device (no CPU fallback).  CPU only; no compute calls."""
import ctypes as C
import os
import re

import liodom_amd as la

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "liodom_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(liodom_[a-z_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    la.build()
    L = C.CDLL(la.lib_path())
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "missing export: " + n
    assert sorted(la.api.EXPORTED_SYMBOLS) == names


def test_params_defaults_mirror_reference():
    p = la.make_params()
    # src/params.cc:40-109
    assert (p.min_range, p.max_range) == (3.0, 75.0)
    assert (p.lidar_type, p.scan_lines, p.scan_regions, p.edges_per_region) == (0, 64, 8, 10)
    assert p.min_points_per_scan == 8 * 10 + 10 and p.local_map_size == 5
    assert (p.save_results, p.use_imu, p.filter_local_map, p.mapping, p.publish_tf) == (0, 0, 0, 0, 1)
    assert p.results_dir == b"~/" and p.fixed_frame == b"odom" and p.base_frame == b"base_link" and p.laser_frame == b""
    # one field per member of liodom::Params (include/liodom/params.h:33-49)
    assert len(la.Params._fields_) == 17
    q = la.make_params(scan_regions=6, edges_per_region=20, prev_frames=12)
    assert q.min_points_per_scan == 130 and q.local_map_size == 12


def test_no_cpu_fallback():
    import subprocess, sys
    # run in a subprocess so that a GPU-less HIP runtime cannot disturb this process
    code = ("import sys; sys.path.insert(0, %r); import liodom_amd as la\n"
            "try:\n    la.Liodom(la.make_params(), la.make_config()); print('CREATED')\n"
            "except la.LiodomError as e:\n    print('REFUSED', e)\n") % ROOT
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env).stdout
    assert "REFUSED" in out and "CREATED" not in out


def test_header_compiles_as_c_and_keeps_deprecated_names(tmp_path):
    """include/liodom_hip.h is a C header: it must compile as C11 and as C++17, and source written against earlier versions of
    it (LIODOM_STATUS_RING_OVERFLOW, liodom_config_t::max_ring_points) must still compile."""
    import subprocess
    src = ('#include "liodom_hip.h"\n'
           'int f(void) { liodom_config_t c; liodom_config_default(&c); c.max_ring_points = 4096;\n'
           '  return (int)(LIODOM_STATUS_RING_OVERFLOW | LIODOM_STATUS_EDGE_OVERFLOW) + c.reserved1 + (int)sizeof(liodom_params_t); }\n')
    for name, cc, std in (("t.c", "gcc", "-std=c11"), ("t.cc", "g++", "-std=c++17")):
        f = tmp_path / name
        f.write_text(src)
        subprocess.check_call([cc, std, "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", str(f), "-o", str(tmp_path / (name + ".o"))])
    # the ctypes mirror has the same size as the C struct
    probe = tmp_path / "sz.c"
    probe.write_text('#include <stdio.h>\n#include "liodom_hip.h"\nint main(void) { printf("%zu %zu\\n", sizeof(liodom_config_t), sizeof(liodom_params_t)); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(probe), "-o", str(exe)])
    a, b = subprocess.check_output([str(exe)], text=True).split()
    assert int(a) == C.sizeof(la.Config) and int(b) == C.sizeof(la.Params)

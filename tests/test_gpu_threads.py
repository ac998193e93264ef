"""The reference's two-thread pipeline through the C-ABI: a FeatureExtractor thread one (or more)
scans ahead of a LaserOdometer thread on the SAME handle (src/liodom_node.cc:89-91; hand-over queue
src/shared_data.cc:64-89).  liodom_extract_edges works on the handle's extraction side,
liodom_odometry_step on its odometry side; run concurrently they must give the same bits as the
serial order extract(k) -> odometry(k) -> extract(k+1) -> ..."""
import ctypes as C
import os
import queue
import subprocess
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CONFIGS = {
    # BASELINE.json configs[0] and configs[2]
    "cfg1_16x900": dict(H=16, W=900, lt=0, R=6, epr=10, P=5, K=30),
    "cfg3_64x1800": dict(H=64, W=1800, lt=0, R=8, epr=10, P=20, K=30),
}


def _handle(c):
    import liodom_amd as la
    return la.Liodom(la.make_params(lidar_type=c["lt"], scan_lines=c["H"], scan_regions=c["R"], edges_per_region=c["epr"],
                                    prev_frames=c["P"]),
                     la.make_config(max_points=c["H"] * c["W"], max_width=c["W"]))


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_extractor_thread_ahead_of_odometer_thread(synth, name):
    c = CONFIGS[name]
    H, W, K = c["H"], c["W"], c["K"]
    cfg = synth.make_cfg(H, W, c["lt"])
    scans = [synth.scan(cfg, 4, k)[0] for k in range(K)]

    # serial order on one thread
    g = _handle(c)
    serial_edges, serial_poses, serial_info = [], [], []
    for k in range(K):
        e = g.extract_edges(scans[k], H, W)
        pose, info = g.odometry_step(e["edges"], stamp=0.1 * k)
        serial_edges.append(e)
        serial_poses.append(pose)
        serial_info.append((info.n_edges, tuple(info.matches), info.lm[0].iterations, info.lm[1].iterations, info.status))
    g.close()

    # two threads, unbounded queue (the reference has no back-pressure either): the extractor runs ahead
    g = _handle(c)
    feats = queue.Queue()
    poses, infos, edges_t, errors = [], [], [], []
    ahead = []                       # how many scans the extractor was ahead whenever the odometer took one
    t_ext, t_odo = [], []            # (start, end) of every call, to show that the two sides really overlap
    done = threading.Event()
    import time

    def extractor():
        try:
            for k in range(K):
                t0 = time.perf_counter()
                e = g.extract_edges(scans[k], H, W)      # ctypes releases the GIL inside the call
                t_ext.append((t0, time.perf_counter()))
                edges_t.append(e)
                feats.put((k, e["edges"]))
        except Exception as ex:       # pragma: no cover
            errors.append(ex)
        finally:
            done.set()
            feats.put(None)

    def odometer():
        try:
            while True:
                item = feats.get()
                if item is None:
                    break
                k, ed = item
                # stay one scan behind: scan k enters odometry once the extractor has finished scan k + 1
                # (or everything), so extract(k + 2) runs while odometry(k) does
                while len(edges_t) < k + 2 and not done.is_set():
                    time.sleep(0.0002)
                ahead.append(len(edges_t) - 1 - k)
                t0 = time.perf_counter()
                pose, info = g.odometry_step(ed, stamp=0.1 * k)
                t_odo.append((t0, time.perf_counter()))
                poses.append(pose)
                infos.append((info.n_edges, tuple(info.matches), info.lm[0].iterations, info.lm[1].iterations, info.status))
        except Exception as ex:       # pragma: no cover
            errors.append(ex)

    ta, tb = threading.Thread(target=extractor), threading.Thread(target=odometer)
    ta.start(); tb.start()
    ta.join(timeout=300); tb.join(timeout=300)
    assert not errors, errors
    assert len(poses) == K
    for k in range(K):
        for key in ("edges", "ring", "idx_in_ring", "src"):
            assert np.array_equal(edges_t[k][key].view(np.uint32) if key == "edges" else edges_t[k][key],
                                  serial_edges[k][key].view(np.uint32) if key == "edges" else serial_edges[k][key]), (k, key)
        assert np.array_equal(poses[k].view(np.uint64), serial_poses[k].view(np.uint64)), k      # bit-equal poses
        assert infos[k] == serial_info[k], k
        assert infos[k][4] == 0                                                                  # no overflow status
    assert min(ahead[:K - 1]) >= 1, "the extractor was not ahead of the odometer"
    overlaps = sum(1 for (a0, a1) in t_ext for (b0, b1) in t_odo if a0 < b1 and b0 < a1)
    assert overlaps >= K // 4, "extract_edges and odometry_step calls of the two threads never overlapped in time"
    # the inspection calls still work after the threaded run
    w, nf = g.window()
    assert nf == min(K, c["P"])
    g.close()


def test_threaded_replay_harness_equals_fused_replay(synth, tmp_path):
    """liodom_amd/host: the C++ mirror's FeatureExtractor::operator() / LaserOdometer::operator() worker
    loops on two std::threads (liodom_replay threads=true) against the fused per-scan call."""
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "liodom_amd", "host", "liodom_replay")
    if not os.path.exists(exe):
        pytest.skip("liodom_replay not built (run __graft_entry__.build())")
    H, W, K = 16, 900, 30
    cfg = synth.make_cfg(H, W, 0)
    scan_dir = tmp_path / "scans"
    scan_dir.mkdir()
    for k in range(K):
        synth.scan(cfg, 0, k)[0].astype(np.float32).tofile(str(scan_dir / ("%06d.bin" % k)))
    outs = {}
    for mode in ("fused", "threads"):
        out_dir = tmp_path / mode
        out_dir.mkdir()
        args = [exe, str(scan_dir), str(out_dir) + "/", "scan_lines=16", "scan_regions=6", "edges_per_region=10", "prev_frames=5"]
        if mode == "threads":
            args.append("threads=true")
        r = subprocess.run(args, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        outs[mode] = np.loadtxt(str(out_dir / "odom.txt")).reshape(K, -1)
        assert np.loadtxt(str(out_dir / "poses.txt")).reshape(-1, 12).shape == (K, 12)
        assert len(np.loadtxt(str(out_dir / "nfeats.txt"))) == K
    a, b = outs["fused"], outs["threads"]
    assert np.array_equal(a[:, :8], b[:, :8])                 # stamp, orientation, position: identical digits
    assert np.allclose(a[1:, 8:], b[1:, 8:], rtol=0, atol=1e-9)   # twist (first row is 0/0 = NaN in both, :125,136)


def _replay(g, K, N, H, W):
    poses, infos = g.replay_resident(0, K, N, H, W, depth=1)
    status = 0
    for i in infos:
        status |= int(i.status)
    return poses, status


@pytest.mark.parametrize("shape", ["two_single_stream_handles", "two_stream_handle_and_single"])
def test_two_handles_on_one_gpu_in_pipelined_replay(synth, shape):
    """The in-kernel cross-stream waits (pipe_wait, the streamed rebuild's pose hand-off, the multi-workgroup solve's
    exchanges) assume that the kernels they wait for can run beside them.  Two handles on ONE GPU, each in pipelined
    replay from its own host thread (two sensors, or two replicas sharing a device), must not disturb each other:
    200 scans, poses bit-equal to the solo runs, no status bit (a wait that gave up would raise
    LIODOM_STATUS_PIPE_TIMEOUT / LM_SYNC_TIMEOUT and fail the replay)."""
    import liodom_amd as la
    H, W, R, epr, P, K = 16, 1800, 8, 20, 10, 200
    N = H * W
    cfg = synth.make_cfg(H, W, 0)
    streams = {sid: [synth.scan(cfg, sid, k)[0] for k in range(K)] for sid in (11, 12, 13)}
    par = la.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P)

    def make(sids):
        g = la.Liodom(par, la.make_config(n_streams=len(sids), max_points=N, max_width=W, pose_log_capacity=K + 8))
        g.alloc_resident(K)
        for s, sid in enumerate(sids):
            for k in range(K):
                g.upload_scan(s, k, streams[sid][k])
        g.sync()
        return g

    layout = [(11,), (12,)] if shape == "two_single_stream_handles" else [(11, 12), (13,)]
    solo = []
    for sids in layout:
        g = make(sids)
        poses, status = _replay(g, K, N, H, W)
        assert status == 0
        solo.append(poses.copy())
        g.close()

    handles = [make(sids) for sids in layout]
    out, errors = [None] * len(handles), []
    start = threading.Barrier(len(handles))

    def run(i):
        try:
            start.wait()
            out[i] = _replay(handles[i], K, N, H, W)
        except Exception as ex:          # noqa: BLE001
            errors.append((i, ex))

    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(handles))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i, g in enumerate(handles):
        poses, status = out[i]
        assert status == 0, "handle %d: status bits 0x%x" % (i, status)
        assert np.array_equal(poses.view(np.uint64), solo[i].view(np.uint64)), "handle %d: poses differ from its solo run" % i
        g.close()


def test_host_fed_replay_equals_resident_replay(synth):
    """liodom_replay_host (scans in page-locked host memory, upload + extraction of scan k+1 beside the odometry of scan k)
    must give the bits of the resident replay."""
    import liodom_amd as la
    H, W, R, epr, P, K = 16, 900, 6, 10, 5, 40
    N = H * W
    cfg = synth.make_cfg(H, W, 0)
    scans = np.stack([synth.scan(cfg, 3, k)[0] for k in range(K)])
    par = la.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P)
    g = la.Liodom(par, la.make_config(max_points=N, max_width=W, pose_log_capacity=2 * K + 8))
    g.alloc_resident(K)
    for k in range(K):
        g.upload_scan(0, k, scans[k])
    ref, _ = g.replay_resident(0, K, N, H, W, depth=1)
    for depth in (1, 0):
        g.reset()
        got, infos = g.replay_host(scans.reshape(K, 1, N, 4), N, H, W, depth=depth)
        assert np.array_equal(got.view(np.uint64), ref.view(np.uint64)), depth
        assert all(int(i.status) == 0 for i in infos)
    g.close()


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_device_handoff_two_threads_equal_serial_order(synth, name):
    """liodom_extract_edges_device / liodom_wait_edges on an extractor thread, liodom_odometry_submit_device /
    liodom_odometry_collect on an odometer thread (src/liodom_node.cc:89-91 with the queue element of
    src/shared_data.cc:64-89 staying in HBM): edges as liodom_extract_edges returns them, poses bit-equal to the serial
    extract -> odometry order with host clouds."""
    import time
    c = CONFIGS[name]
    H, W, K = c["H"], c["W"], c["K"]
    cfg = synth.make_cfg(H, W, c["lt"])
    scans = [synth.scan(cfg, 4, k)[0] for k in range(K)]
    g = _handle(c)
    serial_edges, serial_poses = [], []
    for k in range(K):
        e = g.extract_edges(scans[k], H, W)
        pose, _ = g.odometry_step(e["edges"], stamp=0.1 * k)
        serial_edges.append(e)
        serial_poses.append(pose)
    g.close()

    for depth in (0, 1):
        g = _handle(c)
        tickets = queue.Queue()
        edges_t, poses, infos, errors, busy = [], [], [], [], [0]

        def extractor():
            try:
                for k in range(K):
                    # scans 0, 3, 6, ... are assembled in the handle's page-locked buffer (asynchronous upload), the others
                    # come from pageable memory (copied through the ring)
                    src = scans[k]
                    if k % 3 == 0:
                        buf = g.scan_buffer()
                        buf[:src.shape[0]] = src
                        src = buf[:src.shape[0]]
                    while True:
                        t = g.extract_edges_device(src, H, W)
                        if t is not None:
                            break
                        busy[0] += 1
                        time.sleep(0.0001)
                    edges_t.append(g.wait_edges(t))
                    tickets.put(t)
            except Exception as ex:       # pragma: no cover
                errors.append(ex)
            finally:
                tickets.put(None)

        def odometer():
            try:
                inflight, finished = 0, False
                while not finished or inflight:
                    t = None
                    if not finished and inflight < (2 if depth else 1):
                        try:
                            t = tickets.get(timeout=0.0 if inflight else 5.0)
                        except queue.Empty:
                            t = None
                        else:
                            if t is None:
                                finished = True
                    if t is not None:
                        assert g.odometry_submit_device(t, stamp=0.1 * len(poses))
                        inflight += 1
                        continue
                    if inflight:
                        pose, info = g.odometry_collect()
                        poses.append(pose)
                        infos.append(int(info.status))
                        inflight -= 1
            except Exception as ex:       # pragma: no cover
                errors.append(ex)

        ta, tb = threading.Thread(target=extractor), threading.Thread(target=odometer)
        ta.start(); tb.start()
        ta.join(timeout=300); tb.join(timeout=300)
        assert not errors, errors
        assert len(poses) == K and not any(infos)
        for k in range(K):
            for key in ("edges", "ring", "idx_in_ring", "src"):
                a, b = edges_t[k][key], serial_edges[k][key]
                assert np.array_equal(a.view(np.uint32) if key == "edges" else a, b.view(np.uint32) if key == "edges" else b), (depth, k, key)
            assert np.array_equal(poses[k].view(np.uint64), serial_poses[k].view(np.uint64)), (depth, k)
        w, nf = g.window()
        assert nf == min(K, c["P"])
        g.close()


def test_device_handoff_back_pressure_and_stale_tickets(synth):
    """Three hand-off slots: a fourth extraction without a consumed ticket is refused (LIODOM_ERR_BUSY, nothing enqueued);
    a ticket is good once; the plain entry points refuse while tickets are outstanding; liodom_reset voids tickets."""
    import liodom_amd as la
    c = CONFIGS["cfg1_16x900"]
    H, W = c["H"], c["W"]
    cfg = synth.make_cfg(H, W, 0)
    scans = [synth.scan(cfg, 2, k)[0] for k in range(8)]
    g = _handle(c)
    ref = _handle(c)
    want = [ref.process_scan(scans[k], H, W)[0] for k in range(6)]
    ref.close()
    t = [g.extract_edges_device(scans[k], H, W) for k in range(3)]
    assert all(x is not None for x in t)
    assert g.extract_edges_device(scans[3], H, W) is None              # all three slots filled
    with pytest.raises(la.LiodomError):
        g.process_scan(scans[0], H, W)                                 # plain entry point: tickets outstanding
    p0, _ = g.odometry_step_device(t[0])
    assert np.array_equal(p0.view(np.uint64), want[0].view(np.uint64))
    with pytest.raises(la.LiodomError):
        g.odometry_step_device(t[0])                                   # consumed
    t3 = g.extract_edges_device(scans[3], H, W)                        # slot 0 is free again
    assert t3 is not None and t3.slot == t[0].slot
    got = [g.odometry_step_device(x)[0] for x in (t[1], t[2], t3)]
    for k, p in enumerate(got, start=1):
        assert np.array_equal(p.view(np.uint64), want[k].view(np.uint64)), k
    # two in flight at most
    ta, tb = g.extract_edges_device(scans[4], H, W), g.extract_edges_device(scans[5], H, W)
    assert g.odometry_submit_device(ta) and g.odometry_submit_device(tb)
    tc = g.extract_edges_device(scans[6], H, W)
    assert tc is not None and g.odometry_submit_device(tc) is False    # LIODOM_ERR_BUSY
    for k in (4, 5):
        assert np.array_equal(g.odometry_collect()[0].view(np.uint64), want[k].view(np.uint64)), k
    g.reset()
    with pytest.raises(la.LiodomError):
        g.odometry_step_device(tc)                                     # voided by the reset
    pose, info = g.process_scan(scans[0], H, W)                        # the handle is usable as before
    assert info.status == 0 and np.array_equal(pose.view(np.uint64), want[0].view(np.uint64))
    g.close()


def test_ticket_after_a_replay_needs_a_sync_not_a_retry(synth):
    """A node that prefills with liodom_replay_resident and then switches to the ticket API: while scans of the replay still occupy
    the pipeline edge buffers, liodom_extract_edges_device answers LIODOM_ERR_NEEDS_SYNC (-7) — not LIODOM_ERR_BUSY, whose documented
    reaction is "keep the cloud and retry", which would spin forever — and works after liodom_sync()."""
    import liodom_amd as la
    c = CONFIGS["cfg1_16x900"]
    H, W = c["H"], c["W"]
    cfg = synth.make_cfg(H, W, 0)
    scans = [synth.scan(cfg, 2, k)[0] for k in range(5)]
    g = _handle(c)
    g.alloc_resident(4)
    for k in range(4):
        g.upload_scan(0, k, scans[k])
    poses, _ = g.replay_resident(0, 4, H * W, H, W, depth=1)          # every pose collected; the handle has not been synchronised
    x = np.ascontiguousarray(scans[4], dtype=np.float32)
    t = la.api.EdgeTicket()
    rc = g.L.liodom_extract_edges_device(g.h, 0, x.ctypes.data_as(C.POINTER(C.c_float)), x.size // 4, H, W, C.byref(t))
    assert rc == la.api.ERR_NEEDS_SYNC and rc != la.api.ERR_BUSY
    g.sync()
    tk = g.extract_edges_device(scans[4], H, W)
    assert tk is not None
    p4, info = g.odometry_step_device(tk)
    assert info.status == 0 and info.scan_index == 4
    g.close()


def test_cxx_two_thread_replay_equals_resident_replay(synth):
    """liodom_host_two_thread_replay (two std::threads through the C-ABI, what bench.py's two_thread leg times) gives the
    bits of the resident replay, at both depths, with and without fetching the ~edges clouds."""
    import liodom_amd as la
    H, W, R, epr, P, K = 16, 900, 6, 10, 5, 60
    N = H * W
    cfg = synth.make_cfg(H, W, 0)
    scans = np.stack([synth.scan(cfg, 3, k)[0] for k in range(K)])
    par = la.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P)
    g = la.Liodom(par, la.make_config(max_points=N, max_width=W, pose_log_capacity=2 * K + 8))
    g.alloc_resident(K)
    for k in range(K):
        g.upload_scan(0, k, scans[k])
    ref, infos = g.replay_resident(0, K, N, H, W, depth=1)
    n_edges = sum(int(i.n_edges) for i in infos)
    for depth, fetch, pin in ((1, True, True), (0, True, False), (1, False, True)):
        g.reset()
        got, secs, tot = g.two_thread_replay(scans, N, H, W, timed_from=10, fetch_edges=fetch, depth=depth, pin=pin)
        if not np.array_equal(got.view(np.uint64), ref[:, 0].view(np.uint64)):
            # (diagnostics: the first scan that differs and what the two runs did there)
            d = np.nonzero(np.any(got.view(np.uint64) != ref[:, 0].view(np.uint64), axis=1))[0]
            _, gi = g.pose_log(0, 0, K)
            k = int(d[0])
            what = [(kk, int(infos[kk].n_edges), list(infos[kk].matches), [infos[kk].lm[0].iterations, infos[kk].lm[1].iterations],
                     int(gi[kk].n_edges), list(gi[kk].matches), [gi[kk].lm[0].iterations, gi[kk].lm[1].iterations], int(gi[kk].status),
                     float(np.abs(got[kk] - ref[kk, 0]).max())) for kk in range(max(0, k - 1), min(K, k + 2))]
            raise AssertionError("two-thread replay (depth %d, fetch %s, pin %s) differs from the resident replay on %d scans, first %d: "
                                 "(scan, ref n_edges, matches, iterations | got n_edges, matches, iterations, status, max |dpose|) %s" % (depth, fetch, pin, len(d), k, what))
        assert secs > 0 and tot == (n_edges if fetch else 0)
    g.close()


@pytest.mark.parametrize("shape,scans,distinct,hog", [("vlp16", 10000, 150, "spin"), ("vlp16", 10000, 150, "matmul"), ("hdl64", 3000, 200, "matmul")])
def test_second_process_saturating_the_gpu_never_gives_a_wrong_pose(shape, scans, distinct, hog):
    """tools/soak_two_process.py: a replay with every in-kernel wait active (pipe flags, chain mode, speculative hand-overs, in-launch
    exchanges) while a SECOND PROCESS keeps every CU busy (tools/gpu_hog.hip: bandwidth-bound kernels / 10 ms compute-bound ones).
    Either bit-identical to the solo run, or a clean LIODOM_ERR_HIP and — after liodom_reset(), which enters the safe mode (no
    in-kernel waits) — bit-identical to a solo safe-mode run.  The 64-ring shape beside the compute-bound hog is the one where waits
    really give up: it found the first pass skipping the rebuild's bookkeeping when its own wait had given up (GPU memory fault)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_two_process.py"), shape, str(scans), str(distinct), hog],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, SOAK_BUDGET_S="12"))
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0, (tail, r.stderr[-1500:])
    assert "bit-identical" in tail
    # (the time budget cuts the replay short beside a hog that time-slices the GPU: at least one chunk of the walk must have been compared)
    import re
    m = re.search(r"(\d+) poses compared beside the second process", tail)
    assert m and int(m.group(1)) >= 40, tail


def test_device_handoff_in_safe_mode_uses_events(synth, monkeypatch):
    """LIODOM_SAFE_MODE=1 (what liodom_reset() enters after a timeout): no in-kernel waits — the ticket path then orders the
    odometry behind the extraction with events.  Same bits as liodom_process_scan on a safe-mode handle; agrees with the normal
    mode to rounding (one workgroup per solve: the sums come in another order)."""
    c = CONFIGS["cfg3_64x1800"]
    H, W, K = c["H"], c["W"], 26
    cfg = synth.make_cfg(H, W, 0)
    scans = [synth.scan(cfg, 6, k)[0] for k in range(K)]
    g = _handle(c)
    normal = [g.process_scan(scans[k], H, W)[0] for k in range(K)]
    g.close()
    monkeypatch.setenv("LIODOM_SAFE_MODE", "1")
    g = _handle(c)
    m = g.modes()
    assert m["safe_mode"] == "1" and m["pipe_flags"] == "0" and m["lm_groups"] == "1" and m["early_rebuild"] == "0" and m["knn_overlap"] == "0"
    ref = [g.process_scan(scans[k], H, W)[0] for k in range(K)]
    g.reset()
    got = []
    pending = []
    for k in range(K):
        t = g.extract_edges_device(scans[k], H, W)
        assert t is not None
        pending.append(t)
        if len(pending) == 2:                       # the extractor one scan ahead of the odometer
            got.append(g.odometry_step_device(pending.pop(0))[0])
    got.append(g.odometry_step_device(pending.pop(0))[0])
    g.close()
    for k in range(K):
        assert np.array_equal(got[k].view(np.uint64), ref[k].view(np.uint64)), k
        assert np.max(np.abs(got[k] - normal[k])) < 1e-9, k


def test_chain_mode_hammer_small_shape():
    """tools/chain_hammer.py: fresh handles, 16x900 (short solves: every hand-off of chain mode is at its tightest), resident replay,
    two-thread ticket replay at both depths and per-call pipelined replay, 25 rounds: every pose log must equal the first, bit for
    bit.  Found in round 5: the first pass reading the solve's start point and the edge count from state that finalize_scan was still
    writing, and APPEND reading an edge buffer whose ticket slot the host had already handed to a later scan's extraction."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "chain_hammer.py"), "25", "16x900"], capture_output=True, text=True, timeout=600)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0 and "'resident': 0, 'two_thread': 0, 'percall': 0" in tail, (r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.parametrize("shape,scans", [("hdl64", 240), ("16x900", 400)])
def test_speculative_hand_over_survives_mode_switches(shape, scans):
    """tools/spec_switches.py: the speculative hand-over of the solves' results (kernels_sync.h) with the predictor forced wrong — every
    second pass, every APPEND and every first pass is repaired — across switches into and out of chain mode (per-kernel profiling,
    per-call pipelined path, synchronisations, getters: the repair the host enqueues on its own, chain_flush), and with the model
    predictor: the pose log of a straight replay without speculation, bit for bit."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "spec_switches.py"), shape, str(scans)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "bit-identical" in r.stdout, (r.stdout[-800:], r.stderr[-1500:])


@pytest.mark.parametrize("forced", ["6", "7", "2"])
def test_repeated_pass_is_not_consumed_from_a_stale_l2(forced):
    """A kNN pass that is repeated after a speculative hand-over that was not confirmed is repeated by OTHER workgroups (k_knn_redo,
    k_chain_redo0), on other XCDs than its first edition's; the solve that consumes it has been resident since before either wrote and
    used to rely on "nothing of theirs cached here" — one repair in four left it with the first edition's partial sums or
    correspondences (16 x 900 shape, 256-thread solving workgroups: poses off by 1e-8, no status bit).  LIODOM_SPECULATE=6 / 7 / 2:
    the first / the finalising / both solves hand over as early as possible; six straight chain-mode replays each must equal the
    replay without speculation bit for bit (k_lm_solve invalidates behind its wait when StreamState::spec_redo says so)."""
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "spec_switches.py")
    with tempfile.TemporaryDirectory() as tmp:
        def run(spec, tag):
            f = os.path.join(tmp, tag + ".npy")
            r = subprocess.run([sys.executable, tool, "16x900", "120", f, "straight"], env=dict(os.environ, LIODOM_SPECULATE=spec),
                               capture_output=True, text=True, timeout=300)
            assert r.returncode == 0 and "chain 1" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
            return np.load(f)
        ref = run("0", "ref")
        for i in range(6):
            x = run(forced, "f%d" % i)
            bad = np.nonzero(np.any(x.view(np.uint64) != ref.view(np.uint64), axis=1))[0]
            assert len(bad) == 0, "LIODOM_SPECULATE=%s, run %d: %d scans differ from the replay without speculation, first %s" % (forced, i, len(bad), bad[:5])


@pytest.mark.parametrize("shape,scans", [("hdl64", 2000), ("vlp16", 2000), ("ouster128", 1000)])
def test_schedule_perturbation_leaves_the_pose_log_unchanged(shape, scans):
    """tools/inject_delay.py: a library built with -DLIODOM_INJECT_DELAY (never the product library) delays every publisher of an
    in-kernel hand-off before its store and every waiter after its wait by a pseudo-random 0 .. 20 us — pose / prediction granules,
    done counts and flags, the verdict, pipe flags, chain_release_edges, the solve's exchanges, the appenders' pose.  Chain-mode
    replays with the speculative hand-overs by the model and with every hand-over wrong, three seeds each: pose logs bit-identical
    to the unperturbed run, no status bits.  (Round 5's four faults in these protocols were found by waiting for natural timing to
    produce the bad order; this provokes the orders.)"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "inject_delay.py"), shape, str(scans), "3"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "all perturbed replays bit-identical" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])

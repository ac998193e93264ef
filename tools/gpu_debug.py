"""Diagnostic run on the GPU box: prints detailed GPU-vs-oracle comparisons instead of asserting.
Usage: python tools/gpu_debug.py [section ...]"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liodom_amd as la  # noqa: E402
from liodom_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def mk(H, W, lt=0, R=8, epr=10, P=5, S=1, debug=0, **kw):
    po = orc.make_params(lidar_type=lt, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, knn_mode=1)
    g = la.Liodom(la.make_params(lidar_type=lt, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
                  la.make_config(n_streams=S, max_points=H * W, max_width=W, debug_buffers=debug, **kw))
    return po, g


def sec_extract():
    for (H, W, lt, R, epr) in [(16, 900, 0, 6, 10), (64, 1800, 0, 8, 10), (128, 2048, 1, 8, 10)]:
        cfg = synth.make_cfg(H, W, lt)
        po, g = mk(H, W, lt, R, epr, debug=1)
        x, _ = synth.scan(cfg, 0, 0)
        o = orc.extract(po, x, H, W, want_curv=True)
        t = time.time()
        e = g.extract_edges(x, H, W)
        dt = time.time() - t
        print("extract %dx%d: gpu %d edges, oracle %d edges, %.2f ms (incl. copies)" % (H, W, len(e["ring"]), len(o["ring"]), dt * 1e3))
        cg, offs = g.curvature()
        offs_o, _ = orc.split(po, x, H, W)
        print("  ring offsets equal:", np.array_equal(offs, offs_o))
        if np.array_equal(offs, offs_o):
            co = o["curv"][:len(cg)]
            m = ~(np.isnan(cg) & np.isnan(co))
            bad = (cg[m].view(np.uint64) != co[m].view(np.uint64))
            print("  curvature mismatches: %d of %d" % (bad.sum(), m.sum()))
            if bad.sum():
                i = np.nonzero(m)[0][np.nonzero(bad)[0][:5]]
                print("   first:", i, cg[i], co[i])
        else:
            print("  gpu ring sizes", np.diff(offs)[:20], "\n  orc ring sizes", np.diff(offs_o)[:20])
        n = min(len(e["ring"]), len(o["ring"]))
        same = (e["ring"][:n] == o["ring"][:n]) & (e["idx_in_ring"][:n] == o["idx_in_ring"][:n]) & (e["src"][:n] == o["src"][:n])
        print("  edges identical: %s (first diff at %s)" % (bool(same.all() and len(e["ring"]) == len(o["ring"])), np.nonzero(~same)[0][:3]))
        if not same.all():
            i = np.nonzero(~same)[0][0]
            sl = slice(max(0, i - 3), i + 6)
            print("   gpu", list(zip(e["ring"][sl], e["idx_in_ring"][sl], e["src"][sl])))
            print("   orc", list(zip(o["ring"][sl], o["idx_in_ring"][sl], o["src"][sl])))
        g.close()


def sec_odom(H=16, W=900, lt=0, R=6, epr=10, P=5, K=12):
    cfg = synth.make_cfg(H, W, lt)
    po, g = mk(H, W, lt, R, epr, P)
    od = orc.Odometer(po)
    for k in range(K):
        x, gt = synth.scan(cfg, 0, k)
        o = orc.extract(po, x, H, W)
        pose_o, io = od.step(o["edges"])
        pose_g, ig = g.process_scan(x, H, W)
        dt = np.linalg.norm(pose_g[4:] - pose_o[4:])
        dq = min(np.linalg.norm(pose_g[:4] - pose_o[:4]), np.linalg.norm(pose_g[:4] + pose_o[:4]))
        line = "scan %2d E %d/%d M %d/%d match %s/%s it %s/%s term %s/%s dt %.2e dq %.2e" % (
            k, ig.n_edges, io.n_edges, ig.map_points, io.map_points, list(ig.matches), list(io.matches),
            [ig.lm[0].iterations, ig.lm[1].iterations], [io.lm[0].iterations, io.lm[1].iterations],
            [ig.lm[0].termination, ig.lm[1].termination], [io.lm[0].termination, io.lm[1].termination], dt, dq)
        if k > 0:
            for it in (0, 1):
                vo, ao, bo = od.last_corr(it)
                vg, ag, bg = g.correspondences(it)
                nd = int((vo != vg).sum()) + int(((ao != ag) | (bo != bg))[(vo == 1) & (vg == 1)].sum())
                line += " corrdiff%d=%d" % (it, nd)
            line += " cost %.6g/%.6g" % (ig.lm[1].final_cost, io.lm[1].final_cost)
        wo = od.window()
        wg, nf = g.window()
        line += " win %s/%s maxdiff %.2e" % (wg.shape[0], wo.shape[0], np.abs(wg - wo).max() if wg.shape == wo.shape and len(wo) else -1)
        print(line)
        if ig.status:
            print("  STATUS", ig.status)
    g.close()


def sec_timing(H=64, W=1800, lt=0, R=8, epr=10, P=20, K=60, S=1):
    cfg = synth.make_cfg(H, W, lt)
    po, g = mk(H, W, lt, R, epr, P, S=S, pose_log_capacity=4 * K)
    print("device:", g.device_info())
    g.alloc_resident(K)
    data = [[synth.scan(cfg, d, k)[0] for k in range(K)] for d in range(min(S, 2))]
    for s in range(S):
        for k in range(K):
            g.upload_scan(s, k, data[s % len(data)][k])
    for mode in ("sync", "async", "sync-pipelined", "async-pipelined"):
        g.reset()
        for k in range(10):
            g.process_resident(k, H * W, H, W, readback=True)
        g.sync()
        t = time.time()
        for k in range(10, K):
            g.process_resident(k, H * W, H, W, readback=mode.startswith("sync"),
                               next_slot=(k + 1 if ("pipelined" in mode and k + 1 < K) else -1))
        g.sync()
        dt = time.time() - t
        print("S=%d %s: %.1f us/scan-step, %.0f scans/s aggregate" % (S, mode, dt / (K - 10) * 1e6, S * (K - 10) / dt))
    g.reset()
    g.set_profiling(True)
    for k in range(K):
        g.process_resident(k, H * W, H, W, readback=False)
    st = g.kernel_stats()
    g.set_profiling(False)
    tot = sum(v[1] for v in st.values())
    for name, (n, ms) in st.items():
        if n:
            print("  %-16s %5d launches  avg %8.2f us  total %7.2f ms (%.0f%%)" % (name, n, ms / n * 1e3, ms, 100 * ms / tot))
    _, infos = g.pose_log(0, K - 1, 1)
    print("  last scan: E=%d M=%d matches=%s" % (infos[0].n_edges, infos[0].map_points, list(infos[0].matches)))
    g.close()


SECTIONS = {
    "extract": sec_extract,
    "odom": sec_odom,
    "odom64": lambda: sec_odom(64, 1800, 0, 8, 10, 20, 14),
    "timing": sec_timing,
    "timing_batch": lambda: [sec_timing(S=s, K=40) for s in (8, 64)],
}



def sec_syncprof(H=64, W=1800, R=8, epr=10, P=20, K=60):
    """Kernel durations in per-scan synchronous mode, and host-side timing of each call."""
    cfg = synth.make_cfg(H, W, 0)
    po, g = mk(H, W, 0, R, epr, P, pose_log_capacity=4 * K)
    g.alloc_resident(K)
    for k in range(K):
        g.upload_scan(0, k, synth.scan(cfg, 0, k)[0])
    for k in range(10):
        g.process_resident(k, H * W, H, W, readback=True)
    lat = []
    for k in range(10, K):
        t = time.perf_counter()
        g.process_resident(k, H * W, H, W, readback=True)
        lat.append(time.perf_counter() - t)
    lat = np.array(lat) * 1e6
    print("sync call latency us: min %.1f med %.1f mean %.1f max %.1f" % (lat.min(), np.median(lat), lat.mean(), lat.max()))
    g.reset()
    g.set_profiling(True)
    for k in range(K):
        g.process_resident(k, H * W, H, W, readback=True)
    st = g.kernel_stats()
    g.set_profiling(False)
    for name, (n, ms) in st.items():
        if n:
            print("  [sync] %-16s %5d launches  avg %8.2f us" % (name, n, ms / n * 1e3))
    # sync via stream synchronize instead of polling: enqueue async then sync each scan
    g.reset()
    for k in range(10):
        g.process_resident(k, H * W, H, W, readback=False)
        g.sync()
    t = time.perf_counter()
    for k in range(10, K):
        g.process_resident(k, H * W, H, W, readback=False)
        g.sync()
    print("async+hipStreamSynchronize per scan: %.1f us" % ((time.perf_counter() - t) / (K - 10) * 1e6))
    # sleep between scans (10 Hz-like pacing shrunk to 2 ms) to see idle-clock effects
    g.reset()
    lat = []
    for k in range(K):
        time.sleep(0.002)
        t = time.perf_counter()
        g.process_resident(k, H * W, H, W, readback=True)
        lat.append(time.perf_counter() - t)
    lat = np.array(lat[10:]) * 1e6
    print("paced (2 ms idle between scans) call latency us: min %.1f med %.1f max %.1f" % (lat.min(), np.median(lat), lat.max()))
    g.close()


SECTIONS["syncprof"] = sec_syncprof


def _use_instrumented_library():
    """The phase clocks exist only in a library built with -DLIODOM_INSTRUMENT (the product build carries no instrumentation):
    build that variant (tools/variant_build.sh) if needed and bind this process to it."""
    import subprocess
    import liodom_amd.api as api
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "build", "variants", "libinstrument.so")
    srcs = [os.path.join(root, "liodom_amd", "csrc", f) for f in os.listdir(os.path.join(root, "liodom_amd", "csrc"))]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call([os.path.join(root, "tools", "variant_build.sh"), "instrument", "-DLIODOM_INSTRUMENT"])
    assert api._lib is None, "bind to the instrumented library before anything loads the product library"
    api._LIB = lib
    api.is_stale = lambda: False


def sec_clocks(H=64, W=1800, R=8, epr=10, P=20, K=40):
    import ctypes as C
    os.environ.setdefault("LIODOM_DEBUG_CLOCKS", "65")
    cfg = synth.make_cfg(H, W, 0)
    po, g = mk(H, W, 0, R, epr, P, pose_log_capacity=4 * K)
    g.alloc_resident(K)
    for k in range(K):
        g.upload_scan(0, k, synth.scan(cfg, 0, k)[0])
    serial = os.environ.get("CLK_SERIAL", "0") != "0"        # every scan alone on the GPU (no odometry of the previous scan beside the extraction)
    for k in range(K):
        g.process_resident(k, H * W, H, W, readback=serial)
    g.sync()
    buf = (C.c_ulonglong * 512)()
    g.L.liodom_debug_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    g.L.liodom_debug_clocks(g.h, buf)
    allv = np.array([int(x) if int(x) < 2 ** 63 else int(x) - 2 ** 64 for x in buf], dtype=np.int64)      # (sums of unsigned differences may have wrapped)
    print('k_ring_extract per-ring workgroup durations (us), rings 0..63:', np.round(allv[128:192] / 100.0, 1).tolist())
    print('k_ring_extract (carry re-run rounds, edges) of the rings written last (ring & 31):', [(int(x) & 255, int(x) >> 8) for x in allv[96:128]])
    print('k_knn (both passes): %d queries answered by the Best2 fast path, %d repeated with the exact lists; %d needed a second phase' % (int(allv[256]), int(allv[257]), int(allv[258])))
    print('k_knn second pass: %d queries certified by re-ranking the first pass\'s kept candidates, %d searched' % (int(allv[259]), int(allv[260])))
    print('overlapped second pass: %d queries re-ranked, %d not collected; not certified: %d with fewer than five collected, %d with five; of these: moved > 1 cm %d, > 3 cm %d, guard below the collection radius %d, first-pass fifth distance >= 1 %d' % tuple(int(allv[i]) for i in (261, 262, 264, 265, 266, 269, 267, 268)))
    print('speculative hand-over: %d iterates handed over early; second-pass workgroups: %d confirmed, %d not, %d repeated by k_knn_redo' % tuple(int(allv[i]) for i in (270, 271, 272, 273)))
    print('k_knn query (half-wave) time to selection, 1 us bins:', allv[320:384].tolist())
    print('k_knn candidates streamed per query, bins of 64:', allv[384:448].tolist())
    print('k_knn time (rows: 4 us bins) x candidates (<64, <128, <256, <512, <1024, more | two-phase | exact repeat):')
    print(allv[448:512].reshape(8, 8))
    print('k_knn workgroup-duration histogram (1 us bins, all scans):', allv[192:256].tolist())
    a = allv[:128].reshape(4, 32).copy()
    a[2][30] = 0
    names = {0: ["start", "", "gap bits", "", "", "spec select", "carry resolved", "emitted"],
             1: ["start", "query ready", "hash probed", "phase 1 streamed", "bound + phase 2", "five selected", "nn fetched", "gate + partial sums"],
             2: ["start", "pose ready", "eval0", "begin", "eval1", "upd1", "eval2", "upd2", "eval3", "upd3", "eval4", "upd4", "", "", "", "", "", "", "", "", "loop end", "pose written", "finalized", "", "", "", "", "", "(solving workgroup done)", "appender 0: pose received", "appender 0: point placed"]}
    r2 = a[2]
    print("k_lm_solve (it 1) evaluation done / exchange done (us after kernel start):", [(round((int(allv[64 + 12 + i]) - int(allv[64])) / 100.0, 2), round((int(allv[64 + 4 + 2 * i]) - int(allv[64])) / 100.0, 2)) for i in range(1, 4)])
    print("k_lm_solve (it 1) exchange: local transport %d, XCC %d, evaluations %d" % (int(allv[64 + 27]) // 100, (int(allv[64 + 27]) // 10) % 10, int(allv[64 + 27]) % 10))
    print("k_lm_solve begin phase: controller lm_begin done at %.2f us; evaluator wave 1: start %.2f, compacted %.2f (us after kernel start)" % tuple((r2[i] - r2[0]) / 100.0 for i in (23, 24, 25)))
    for k, kn in ((0, "k_ring_extract (ring 40)"), (1, "k_knn (block 5, it 0)"), (2, "k_lm_solve (it 1)")):
        row = a[k]
        t0 = row[0]
        print(kn)
        prev = t0
        for i, nm in enumerate(names[k]):
            if i == 0 or row[i] == 0 or not nm:
                continue
            print("   %-22s +%7.2f us  (at %7.2f)" % (nm, (row[i] - prev) / 100.0, (row[i] - t0) / 100.0))
            prev = row[i]
    g.close()


def sec_ovclocks(H=64, W=1800, R=8, epr=10, P=20, K=int(os.environ.get("OV_SCANS", "40"))):
    """Wall-clock stamps around the overlapped second kNN pass of the last scan (instrumented build, debug bit 7)."""
    import ctypes as C
    os.environ.setdefault("LIODOM_DEBUG_CLOCKS", "128")
    cfg = synth.make_cfg(H, W, 0)
    po, g = mk(H, W, 0, R, epr, P, pose_log_capacity=4 * K)
    print("modes:", g.modes())
    g.alloc_resident(K)
    for k in range(K):
        g.upload_scan(0, k, synth.scan(cfg, 0, k)[0])
    serial = os.environ.get("KNN_SERIAL", "0") != "0"
    if os.environ.get("OV_REPLAY", "0") != "0":          # the bench's loop (liodom_replay_resident, depth 1) instead of one call per scan
        g.replay_resident(0, K, H * W, H, W, depth=1)
    else:
        for k in range(K):
            g.process_resident(k, H * W, H, W, readback=serial)
    g.sync()
    buf = (C.c_ulonglong * 512)()
    g.L.liodom_debug_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    g.L.liodom_debug_clocks(g.h, buf)
    allv = np.array([int(x) if int(x) < 2 ** 63 else int(x) - 2 ** 64 for x in buf], dtype=np.int64)      # (sums of unsigned differences may have wrapped)
    print("overlapped pass: pose publication -> seen by a workgroup, 0.25 us bins (all scans):", allv[320:384].tolist())
    print("overlapped pass: pose seen -> done flag raised, per workgroup with queries, 0.5 us bins (all scans):", allv[256:320].tolist())
    print("first pass: workgroup end relative to the start of workgroup 0, 0.5 us bins (all scans):", allv[384:448].tolist())
    a = allv[448:480]
    names = ["solve0 wg0 start", "solve0 pose published", "solve0 wg0 end", "solve1 wg0 start", "solve1 wait done", "solve1 wg0 end", "gate start", "gate saw flag",
             "knn1 wg0 start", "knn1 wg0 past flag", "knn1 wg0 pose seen", "knn1 wg0 end", "knn1 wgN start", "knn1 wgN past flag", "knn1 wgN pose seen", "knn1 wgN end",
             "knn0 wg0 start", "knn0 wgN end", "solve0 partial sums next", "solve0 iterate left early", "", "", "", "", "", "", "append wg0 end", "knn0 wg0 past its waits", "append wg0 pose seen"]
    sm = allv[480:496].astype(np.float64)
    if sm[5] > 0:
        print("phase means over %d scans (us): first pass %.2f | first solve %.2f | second pass's tail (pose published -> finalising solve has its sums' inputs) %.2f | finalising solve %.2f | APPEND %.2f | period %.2f; iterate handed over early in %d scans" % (
            int(sm[5]), sm[0] / sm[5] / 100, sm[1] / sm[5] / 100, sm[2] / sm[5] / 100, sm[3] / sm[5] / 100, sm[7] / sm[5] / 100, sm[4] / sm[5] / 100, int(sm[6])))
    if sm[5] > 0:
        print("   second pass: last workgroup done %.2f us after the (confirmed) pose's publication; the finalising solve has seen the count %.2f us later; early hand-overs that were confirmed (%d scans): %.2f us before the confirmation, last workgroup %.2f us after the hand-over, tail %.2f us; not confirmed (%d scans): tail %.2f us" % (
            sm[8] / sm[5] / 100, sm[9] / sm[5] / 100, int(sm[12]), sm[10] / max(sm[12], 1) / 100, sm[11] / max(sm[12], 1) / 100, sm[13] / max(sm[12], 1) / 100, int(sm[14]), sm[15] / max(sm[14], 1) / 100))
    t0 = a[16] if a[16] else a[0]
    order = sorted(range(len(names)), key=lambda i: a[i])
    for i in order:
        if a[i]:
            print("   %-26s %8.2f us" % (names[i], (a[i] - t0) / 100.0))
    g.close()


def sec_knntimes(H=64, W=1800, R=8, epr=10, P=20, K=40):
    """Per-query phase times of k_knn for the last scan of a pipelined replay (LIODOM_DEBUG_CLOCKS)."""
    import ctypes as C
    os.environ.setdefault("LIODOM_DEBUG_CLOCKS", "1")
    cfg = synth.make_cfg(H, W, 0)
    po, g = mk(H, W, 0, R, epr, P, pose_log_capacity=4 * K)
    g.alloc_resident(K)
    for k in range(K):
        g.upload_scan(0, k, synth.scan(cfg, 0, k)[0])
    serial = os.environ.get("KNN_SERIAL", "0") != "0"      # every scan alone on the GPU (no extraction of the next scan beside it)
    print("mode:", "serial (pose read back before the next scan is submitted)" if serial else "asynchronous replay (next scan's extraction overlaps)")
    for k in range(K):
        g.process_resident(k, H * W, H, W, readback=serial)
    g.sync()
    cap = C.c_int(0)
    g.L.liodom_debug_knn_times.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    g.L.liodom_debug_knn_times(g.h, None, C.byref(cap))
    buf = np.zeros((2, cap.value, 12), dtype=np.uint32)
    assert g.L.liodom_debug_knn_times(g.h, buf.ctypes.data_as(C.c_void_p), C.byref(cap)) == 0
    E = int(np.count_nonzero(buf[0, :, 7] | buf[0, :, 6]))
    names = ["query ready", "probed", "phase 1", "phase 2", "selected", "barrier", "end"]
    for it in (0, 1):
        t = buf[it, :E, 1:8].astype(np.float64) / 100.0
        x = buf[it, :E, 8:]
        n = buf[it, :E, 0] & 0x3FFFFFFF
        two = (buf[it, :E, 0] >> 30) & 1
        d = np.diff(np.concatenate([np.zeros((E, 1)), t], axis=1), axis=1)
        print("pass %d: %d queries; mean phase durations (us): %s" % (it, E, ", ".join("%s %.2f" % (nm, x) for nm, x in zip(names, d.mean(axis=0)))))
        sel = t[:, 4]
        order = np.argsort(-sel)
        print("  selection time percentiles (us): 50%% %.2f  90%% %.2f  99%% %.2f  max %.2f" % tuple(np.percentile(sel, [50, 90, 99, 100])))
        for q in order[:12]:
            print("  slow query %4d: n=%5d two-phase=%d  %s | phase-1 stream: big part %.2f us (%d cells, %d rounds), flat part %.2f us (%d rounds, T=%d)" % (
                q, n[q], two[q], "  ".join("%s +%.2f" % (nm, xx) for nm, xx in zip(names, d[q])), x[q, 0] / 100.0, x[q, 2] & 255, (x[q, 2] >> 8) & 255, x[q, 1] / 100.0, (x[q, 2] >> 16) & 15, x[q, 2] >> 20))
        flat_t = x[:, 1] / 100.0
        rounds = (x[:, 2] >> 16) & 15
        nseg = x[:, 3]
        pair_rounds = np.maximum(rounds[0::2][:E // 2], rounds[1::2][:E // 2])
        pair_nseg = np.maximum(nseg[0::2][:E // 2], nseg[1::2][:E // 2])
        pair_t = flat_t[0::2][:E // 2]
        for r in (1, 2, 3):
            for lo, hi in ((1, 2), (3, 4), (5, 8), (9, 28)):
                msk = (pair_rounds == r) & (pair_nseg >= lo) & (pair_nseg <= hi)
                if msk.sum():
                    print("   phase-1 flat part, %d round(s), %2d..%2d segments: %4d wave(s), mean %.2f us, 90%% %.2f, max %.2f" % (r, lo, hi, msk.sum(), pair_t[msk].mean(), np.percentile(pair_t[msk], 90), pair_t[msk].max()))
        wg = sel[:E - E % 8].reshape(-1, 8)
        print("  workgroup (8 queries) max selection time: mean %.2f us, max %.2f; mean of all queries %.2f" % (wg.max(axis=1).mean(), wg.max(), sel.mean()))
    g.close()


SECTIONS["knntimes"] = sec_knntimes
def sec_hbclocks(H=64, W=1800, R=8, epr=10, P=20, K=36):
    """Phase stamps of k_hash_build (stream 0's workgroup) on a lock-step batch (instrumented build, debug bit 7)."""
    import ctypes as C
    os.environ.setdefault("LIODOM_DEBUG_CLOCKS", "128")
    S = int(os.environ.get("HB_STREAMS", "256"))
    cfg = synth.make_cfg(H, W, 0)
    import liodom_amd as la
    g = la.Liodom(la.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
                  la.make_config(n_streams=S, max_points=H * W, max_width=W, pose_log_capacity=K + 8))
    g.alloc_resident(K)
    scans = [synth.scan(cfg, 0, k)[0] for k in range(K)]
    for s in range(S):
        for k in range(K):
            g.upload_scan(s, k, scans[k])
    for k in range(K):
        g.process_resident(k, H * W, H, W, readback=False, next_slot=(k + 1 if k + 1 < K else -1))
    g.sync()
    buf = (C.c_ulonglong * 512)()
    g.L.liodom_debug_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    g.L.liodom_debug_clocks(g.h, buf)
    a = np.array(list(buf), dtype=np.int64)[448 + 19:448 + 25]
    names = ["start", "table initialised", "insert + count done", "prefix done", "scatter done (after barrier)", "table published"]
    for i in range(1, 6):
        print("   %-30s +%7.2f us  (at %7.2f)" % (names[i], (a[i] - a[i - 1]) / 100.0, (a[i] - a[0]) / 100.0))
    g.close()


def sec_knn8phases(H=64, W=1800, R=8, epr=10, P=20, K=30):
    """Shader cycles per phase of k_knn8 on a lock-step batch, summed over the working waves (instrumented build, debug bit 8)."""
    import ctypes as C
    os.environ.setdefault("LIODOM_DEBUG_CLOCKS", "256")
    S = int(os.environ.get("HB_STREAMS", "256"))
    cfg = synth.make_cfg(H, W, 0)
    import liodom_amd as la
    g = la.Liodom(la.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
                  la.make_config(n_streams=S, max_points=H * W, max_width=W, pose_log_capacity=K + 8))
    g.alloc_resident(K)
    scans = [synth.scan(cfg, 0, k)[0] for k in range(K)]
    for s in range(S):
        for k in range(K):
            g.upload_scan(s, k, scans[k])
    for k in range(K):
        g.process_resident(k, H * W, H, W, readback=False, next_slot=(k + 1 if k + 1 < K else -1))
    g.sync()
    buf = (C.c_ulonglong * 512)()
    g.L.liodom_debug_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    g.L.liodom_debug_clocks(g.h, buf)
    a = np.array(list(buf), dtype=np.float64)
    names = ["query + own probe + sort", "second pass: re-rank + certificate", "own cell", "bound + neighbour cells", "selection", "save for the second pass", "neighbours fetched, records written"]
    for it in (0, 1):
        row = a[64 + 8 * it:64 + 8 * it + 7]
        tot = row.sum()
        print("k_knn8<%d>: %s" % (it, ", ".join("%s %.1f %%" % (n, 100.0 * x / max(tot, 1.0)) for n, x in zip(names, row))))
    print("second pass: %d wave-iterations, %d with at least one searching query, %d searching queries (%.1f %% of %d)" % (a[64 + 23], a[64 + 7], a[64 + 15], 100.0 * a[64 + 15] / max(8 * a[64 + 23], 1), 8 * a[64 + 23]))
    g.close()


SECTIONS["knn8phases"] = sec_knn8phases
SECTIONS["hbclocks"] = sec_hbclocks
SECTIONS["ovclocks"] = sec_ovclocks


SECTIONS["clocks"] = sec_clocks
SECTIONS["timing_ouster"] = lambda: sec_timing(H=128, W=2048, lt=1, R=8, epr=10, P=30, K=45)
SECTIONS["timing_vlp16"] = lambda: sec_timing(H=16, W=1800, lt=0, R=8, epr=20, P=10, K=60)


def sec_long(H=64, W=1800, R=8, epr=10, P=20, K=220):
    cfg = synth.make_cfg(H, W, 0)
    po, g = mk(H, W, 0, R, epr, P)
    od = orc.Odometer(po)
    shown = 0
    for k in range(K):
        x, gt = synth.scan(cfg, 0, k)
        o = orc.extract(po, x, H, W)
        pose_o, io = od.step(o["edges"])
        pose_g, ig = g.process_scan(x, H, W)
        dt = np.linalg.norm(pose_g[4:] - pose_o[4:])
        eg = np.linalg.norm(pose_g[4:] - gt[4:])
        eo = np.linalg.norm(pose_o[4:] - gt[4:])
        flag = dt > 1e-9 or ig.n_edges != io.n_edges or list(ig.matches) != list(io.matches)
        if (flag and shown < 25) or k % 20 == 0:
            shown += flag
            nd = []
            for it in (0, 1):
                vo, ao, bo = od.last_corr(it)
                vg, ag, bg = g.correspondences(it)
                if len(vo) == len(vg):
                    nd.append(int((vo != vg).sum()) + int(((ao != ag) | (bo != bg))[(vo == 1) & (vg == 1)].sum()))
                else:
                    nd.append(-1)
            print("scan %3d E %d/%d M %d/%d match %s/%s it %s/%s term %s/%s acc %s/%s dt %.2e err_gt gpu %.3f orc %.3f corrdiff %s cost %.6g/%.6g st %d" % (
                k, ig.n_edges, io.n_edges, ig.map_points, io.map_points, list(ig.matches), list(io.matches),
                [ig.lm[0].iterations, ig.lm[1].iterations], [io.lm[0].iterations, io.lm[1].iterations],
                [ig.lm[0].termination, ig.lm[1].termination], [io.lm[0].termination, io.lm[1].termination],
                [ig.lm[0].accepted, ig.lm[1].accepted], [io.lm[0].accepted, io.lm[1].accepted],
                dt, eg, eo, nd, ig.lm[1].final_cost, io.lm[1].final_cost, ig.status), flush=True)
    g.close()


SECTIONS["long"] = sec_long


if __name__ == "__main__":
    names = sys.argv[1:] or ["extract", "odom", "odom64", "timing"]
    if any(n in ("clocks", "knntimes", "ovclocks", "hbclocks", "knn8phases") for n in names):
        _use_instrumented_library()
    for n in names:
        print("=" * 20, n)
        try:
            SECTIONS[n]()
        except Exception:
            traceback.print_exc()
        sys.stdout.flush()




#!/usr/bin/env python3
"""bench.py — LiODOM hot path on MI355X: scans/sec on the BASELINE.json headline workload.

    python bench.py --gpus N --steps K --warmup W [--workload hdl64|vlp16|ouster128]

A "step" is one pass of the hot path over one batch of synthetic input: one scan of one stream per
GPU (edge extraction -> 2 x [correspondences + pose solve] -> window update), with the pose read back
every scan as the ROS node publishes it.  All scans are resident in HBM before the timed region, and
the sliding window is pre-filled to prev_frames frames (untimed, before the W warm-up steps), so every
timed step runs in the steady state whatever K and W are.  The path does not shard (each scan depends
on the previous pose and window), so N > 1 GPUs run N independent replayed streams ("replicas only",
no collective on the data path); torch.distributed (gloo) is used only for the barrier / max-over-ranks
timing and for collecting each replica's parity.

Output: ONE JSON line on rank 0 with metric/value/... plus
  roofline      dominant kernel of the timed workload: algorithmic bytes / HIP-event duration
  cpu_baseline  the CPU oracle ("port") on a bounded sample of the same scans, timed on this host:
                1 thread, and the reference's own thread policy
  parity        GPU-vs-oracle pose difference, per replica (worst over ranks)
  roofline_8d   the same timed leg scored per STAGE against SURVEY.md §8(d)'s bytes (extract = 16 N + 24 E for classify +
                scatter + extract + compact together, ...)
  host_fed / two_thread   the drop-in-shaped rates: scans arriving in pinned HOST memory (liodom_replay_host: upload and
                extraction of scan k+1 overlap the odometry of scan k), and an extractor thread + an odometer thread
                through liodom_extract_edges / liodom_odometry_step as INTEGRATION.md §2 prescribes (never `value`)
  batched       lock-step multi-stream run on one GPU (throughput mode) with its own roofline
Exit status 3 if any replica's parity check fails.
"""
import argparse
import ctypes
import json
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # OpenMP threads (generator, CPU baseline) must not spin during GPU timing
os.environ.setdefault("OMP_PROC_BIND", "false")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)

WORKLOADS = {
    # BASELINE.json configs[2] (headline): HDL-64-shape 64x1800, scan_regions=8, prev_frames=20
    "hdl64": dict(H=64, W=1800, lidar_type=0, R=8, epr=10, P=20,
                  name="HDL-64-shape 64x1800 synthetic stream, scan_regions=8, edges_per_region=10, prev_frames=20"),
    # the headline shape with ragged input (dropped points under load: ~25 % NaN no-returns, rings of unequal length, three rings
    # below min_points_per_scan; liodom_amd.synth.ragged) — not a BASELINE config, a robustness leg
    "hdl64_ragged": dict(H=64, W=1800, lidar_type=0, R=8, epr=10, P=20, ragged=True,
                         name="HDL-64-shape 64x1800 synthetic stream with ragged returns (~25 % NaN no-returns, unequal rings, 3 rings below "
                              "min_points_per_scan), scan_regions=8, edges_per_region=10, prev_frames=20"),
    # configs[1]: VLP-16-shape 16x1800, scan_regions=8, edges_per_region=20, prev_frames=10
    "vlp16": dict(H=16, W=1800, lidar_type=0, R=8, epr=20, P=10,
                  name="VLP-16-shape 16x1800 synthetic stream, scan_regions=8, edges_per_region=20, prev_frames=10"),
    # configs[3]: Ouster-128-shape 128x2048 (liodom_ouster.launch: lidar_type=1, R=8, epr=10), prev_frames=30
    "ouster128": dict(H=128, W=2048, lidar_type=1, R=8, epr=10, P=30,
                      name="Ouster-128-shape 128x2048 synthetic stream (liodom_ouster.launch params), prev_frames=30"),
}


def algorithmic_bytes(kernel, N, E, M, C, evals, streamed=False):
    """SURVEY.md §8(d) per-scan figures, per launch of `kernel` for one stream.  streamed: handles with <= 4 streams
    build the next cell hash with extra workgroups of the two k_lm_solve launches of a scan (32 B per window point per
    scan, half on each launch) instead of k_window_insert / k_hash_alloc / k_hash_scatter."""
    if kernel == "k_classify":
        return 17.0 * N                       # 16 B/point read, 1 id byte written
    if kernel == "k_ring_scatter":
        return 37.0 * N                       # 16 B + id read, 16 B + 4 B source index written (the one-pass splits — k_ring_split, k_ring_split_lb — move 36 N: scored on the same figure)
    if kernel == "k_ring_extract":
        return 16.0 * N + 24.0 * E            # extract: 16 B/point read, 24 B/edge written
    if kernel == "k_knn":
        return 16.0 * (M + E) + 28.0 * E      # one kNN pass: map + queries read, (a, b, flag) written
    if kernel == "k_lm_solve":
        return (36.0 * C + 224.0) * max(evals, 1.0) + (16.0 * M if streamed else 0.0)   # per residual/Jacobian evaluation (+ half of the streamed rebuild)
    if kernel in ("k_window_insert", "k_hash_scatter", "k_hash_alloc", "k_hash_clear", "k_hash_build"):
        return 32.0 * M                       # window / hash rebuild
    if kernel == "k_compact_edges":
        return 32.0 * E
    return 0.0


def _traffic_profiles(workload):
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic_%s.json" % workload)))


def traffic_source(workload):
    """Where `roofline.traffic` comes from: the committed profile's file name and how its PMC passes were run."""
    files = _traffic_profiles(workload)
    if not files:
        return None
    return {"profile": os.path.relpath(files[-1], ROOT),
            "collected_with": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/profile_round.sh -> tools/pmc_summary.py); "
                              "FETCH_SIZE doubled (gfx950 correction of MI355X_MICROARCH.md)",
            "mode_note": "--pmc serialises kernels across HIP streams, so the PMC passes run the EVENT path of the handle (LIODOM_PIPE_FLAGS=0: no "
                         "in-kernel waits between streams, second kNN pass behind the first solve instead of beside it); same kernels and bytes per "
                         "launch as the timed leg, different overlap — a committed profile, not a counter read in this run"}


def measured_traffic(kernel, n_streams, workload):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC profile of THIS workload
    (profiles/*_pmc_traffic_<workload>.json, written by tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE passes;
    FETCH_SIZE doubled as the microarch guide prescribes for gfx950).  Rows are selected by launch shape — the profile
    keeps the single-stream run and the lock-step run (its stream count recorded) apart, launch-weighted over the grids a
    kernel is launched with — never by "smallest grid".  PMC counters cannot be collected from inside this process; null
    when no profile of this workload is committed."""
    files = _traffic_profiles(workload)
    if not files:
        return None
    try:
        prof = json.load(open(files[-1]))
        if prof.get("workload") != workload:
            return None
        if n_streams == 1:
            row = prof.get("single", {}).get(kernel)
            return int(row["hbm_bytes_per_launch"]) if row else None
        for b in prof.get("batched_groups", [prof.get("batched", {})]):
            if b.get("streams") != n_streams:                  # a measurement at THIS stream count only: nothing is scaled
                continue
            rows = b.get("kernels", {})
            # The HIP-event slots of the library (k_knn, k_ring_scatter, k_hash_build) cover several kernels on lock-step batches since
            # round 6: per launch of the slot = the kernels' bytes per launch, weighted by how often each runs per launch of the slot
            parts = BATCH_KERNELS.get(kernel)
            if parts and any(k in rows for k, _ in parts):
                return int(sum(w * rows[k]["hbm_bytes_per_launch"] for k, w in parts if k in rows))
            row = rows.get(kernel)
            if row:
                return int(row["hbm_bytes_per_launch"])
        return None
    except Exception:
        return None


# lock-step batches (>= 16 streams), round 6: what runs inside the library's per-kernel timing slots, and how often per launch of the slot
BATCH_KERNELS = {
    "k_knn": (("k_knn8", 1.0), ("k_knn8_exact", 1.0), ("k_line_gate", 1.0)),                 # one pass: search + exact lists of the uncertain + line gates
    "k_ring_scatter": (("k_ring_split_lb", 1.0), ("k_ring_split_fix", 1.0)),                  # the one-pass ring split (+ its idle repair launch)
    "k_hash_build": (("k_hash_append", 0.75), ("k_hash_build", 0.25)),                       # three appends per rebuild (kHbPeriod = 4)
}


STAGES_8D = {
    # stage -> (kernels whose time counts, bytes per scan)
    "extract": (("k_classify", "k_ring_scatter", "k_ring_extract", "k_compact_edges"), lambda N, E, M, C, ev: 16.0 * N + 24.0 * E),
    "knn (2 passes)": (("k_knn",), lambda N, E, M, C, ev: 2.0 * (16.0 * (M + E) + 28.0 * E)),
    "solve + window/hash rebuild + append": (("k_lm_solve", "k_window_insert", "k_hash_alloc", "k_hash_scatter", "k_hash_build", "k_hash_clear"),
                                             lambda N, E, M, C, ev: 2.0 * (36.0 * C + 224.0) * max(ev, 1.0) + 32.0 * M + 32.0 * E),
}


def roofline_8d(stats, n_streams, N, E, M, C, evals):
    """Stages scored against SURVEY.md §8(d)'s algorithmic bytes per scan (all streams of a step together): summed
    HIP-event kernel time of the stage per step vs its bytes — e.g. extract = 16 N + 24 E for the three extraction
    passes + compaction together, whatever traffic the implementation's extra passes cause."""
    stats = {k: v for k, v in stats.items() if v[0]}
    steps = max(1, stats.get("k_ring_extract", stats.get("k_classify", (1, 0)))[0])      # (organised clouds have no k_classify launch: k_row_compact, booked as k_ring_scatter)
    out, tot_b, tot_us = {}, 0.0, 0.0
    for name, (kernels, fn) in STAGES_8D.items():
        us = sum(stats[k][1] for k in kernels if k in stats) / steps * 1e3
        by = fn(N, E, M, C, evals) * n_streams
        tot_b += by
        tot_us += us
        out[name] = {"us_per_step": round(us, 2), "bytes_per_step": int(by),
                     "achieved_GBs": round(by / (us * 1e-6) / 1e9, 2) if us > 0 else 0.0,
                     "frac": round(by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5) if us > 0 else 0.0}
    out["all stages"] = {"us_per_step": round(tot_us, 2), "bytes_per_step": int(tot_b),
                         "achieved_GBs": round(tot_b / (tot_us * 1e-6) / 1e9, 2) if tot_us > 0 else 0.0,
                         "frac": round(tot_b / (tot_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5) if tot_us > 0 else 0.0,
                         "note": "summed kernel time (extraction overlaps odometry on a second stream in the timed leg)"}
    return out


def roofline_from_stats(stats, n_streams, N, E, M, C, evals, workload="hdl64"):
    """stats: {kernel: (launches, total_ms)} from HIP events on the handle's stream."""
    stats = {k: v for k, v in stats.items() if v[0]}
    streamed = n_streams <= 4 and "k_window_insert" not in stats and "k_hash_build" not in stats
    tot = sum(ms for _, ms in stats.values()) or 1.0
    name, (launches, ms) = max(stats.items(), key=lambda kv: kv[1][1])
    avg_s = ms / max(launches, 1) * 1e-3
    by = algorithmic_bytes(name, N, E, M, C, evals, streamed) * n_streams
    achieved = by / avg_s / 1e9 if avg_s > 0 else 0.0
    per_scan_bytes = sum(algorithmic_bytes(k, N, E, M, C, evals, streamed) * v[0] for k, v in stats.items())   # all launches
    scans = max(1, stats.get("k_ring_extract", stats.get("k_classify", (1, 0)))[0])
    return {
        "bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": measured_traffic(name, n_streams, workload),
        "traffic_source": traffic_source(workload),
        "kernel_time_source": "HIP events around every launch, on a replay of the same K steps with per-kernel profiling on (one stream, the four-launch "
                              "chain k_knn / k_lm_solve / k_knn / k_lm_solve with the streamed rebuild's workgroups inside the solve launches).  The timed "
                              "leg of one-stream handles runs chain mode: there a solve launch is resident while the kNN pass before it still runs, so its "
                              "rocprofv3 duration includes that wait (profiles/*_bench_kernel_trace.txt lists both)",
        "avg_kernel_us": round(avg_s * 1e6, 2), "algorithmic_bytes_per_launch": int(by),
        "share_of_gpu_time": round(ms / tot, 3),
        "per_kernel_us": {k: round(v[1] / max(v[0], 1) * 1e3, 2) for k, v in stats.items()},
        # every kernel against the same roof: algorithmic GB/s and fraction of the 8 TB/s peak
        "per_kernel_frac": {k: round(algorithmic_bytes(k, N, E, M, C, evals, streamed) * n_streams / (v[1] / v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                            for k, v in stats.items() if v[1] > 0 and algorithmic_bytes(k, N, E, M, C, evals, streamed) > 0},
        # all kernels of a step together: algorithmic bytes of one step / summed kernel time of one step
        "end_to_end_frac_of_kernel_time": round(per_scan_bytes / scans * n_streams / (tot / scans * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        # the batch kernels are not HBM kernels: their second roofline, VALU issue (SQ_ACTIVE_INST_VALU quad-cycles / SIMD cycles of the
        # launch, 256 lock-step streams) — from the committed PMC profiles, not a counter read in this run
        **({"valu_issue": {"k_knn8 first pass (256 streams)": 0.60, "k_knn8 both passes (64 streams)": 0.49, "k_ring_split_lb (64 streams)": 0.63, "k_ring_extract (64 streams)": 0.49,
                           "source": "profiles/r06_knn_budget.txt section 3 (256 streams), profiles/r06_f_sq.txt (64 streams: 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x SQ_BUSY_CYCLES / 32))",
                           "note": "fraction of the launch's SIMD cycles that issue a VALU instruction; k_knn8's length is set by a chain of ~20 dependent "
                                   "memory round trips per workgroup at 76 % mean wave occupancy (profiles/r06_knn_budget.txt section 5)"}} if n_streams >= 16 else {}),
    }


def pose_errors(poses_a, poses_b):
    dt = np.linalg.norm(poses_a[:, 4:] - poses_b[:, 4:], axis=1)
    dots = np.abs(np.sum(poses_a[:, :4] * poses_b[:, :4], axis=1))
    dr = 2.0 * np.arccos(np.minimum(1.0, dots))
    return dt, dr


def run_oracle(orc, wl, scans, threads=(1, 1), time_from=0):
    """The CPU oracle on `scans`; returns (poses, seconds spent on scans[time_from:])."""
    po = orc.make_params(lidar_type=wl["lidar_type"], scan_lines=wl["H"], scan_regions=wl["R"], edges_per_region=wl["epr"],
                         prev_frames=wl["P"], knn_mode=1)
    orc.set_threads(*threads)
    od = orc.Odometer(po)
    poses = np.zeros((len(scans), 7))
    tc = 0.0
    for k, x in enumerate(scans):
        ts = time.perf_counter()
        e = orc.extract(po, x, wl["H"], wl["W"])
        pose, _ = od.step(e["edges"])
        te = time.perf_counter()
        if k >= time_from:
            tc += te - ts
        poses[k] = pose
    od.close()
    orc.set_threads(1, 1)
    return poses, tc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="hdl64", choices=sorted(WORKLOADS))
    ap.add_argument("--batched-streams", type=int, default=256, help="lock-step streams of the throughput leg (0 = skip)")
    ap.add_argument("--batched-data-streams", type=int, default=8, help="distinct synthetic streams replayed by the batched leg")
    ap.add_argument("--repeats", type=int, default=9,
                    help="the K-step timed region is measured this many times (each after a reset + untimed pre-fill + W warm-up steps, "
                         "one more discarded first); `value` is the median")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=100, help="timed scans of the CPU baseline sample (bounded)")
    ap.add_argument("--parity-scans", type=int, default=0, help="scans compared with the oracle per replica (0 = prefill + warm-up + min(steps, 100))")
    args = ap.parse_args()

    # The process that drives the GPU on ONE core, from before the HIP runtime starts its own threads (they inherit the mask) — what
    # `taskset -c <core> python bench.py` does.  Measured on the 2 x 64-core host of the MI355X boxes (tools/replay_trace.py,
    # PIN_CPUS): the host needs 26 us to enqueue the launches of one chain-mode scan that way, 42 us when the scheduler may move the
    # threads within the GPU's NUMA node and 54 us within the other node — against a scan period of 70 us: 14.4k scans/s instead of
    # 13.0-13.9k depending on where the process happened to land.  The two-thread leg and the CPU baselines get the full mask
    # back.  LIODOM_BENCH_PIN=0: no pinning beyond the GPU's NUMA node.
    orig_affinity = os.sched_getaffinity(0)
    pin_core = None
    pin_source = "none"
    wide_affinity = orig_affinity            # what the multi-threaded legs (two_thread) get back: the GPU's socket if known
    if os.environ.get("LIODOM_BENCH_PIN", "1") != "0":
        try:
            # the core comes from the GPU's own socket (sysfs, before HIP loads): with `sorted(affinity)[LOCAL_RANK]` the replicas of
            # an 8-GPU node all sat on socket 0, half of them driving their GPU across the socket link (54 us of enqueue per scan
            # instead of 26: round-5 measurement above)
            from liodom_amd.replicas import gpu_local_cpus, choose_core
            lr = int(os.environ.get("LOCAL_RANK", "0"))
            local = gpu_local_cpus()
            pin_core = choose_core(lr % len(local), local, orig_affinity) if local else None
            pin_source = "gpu local_cpulist (sysfs)"
            if pin_core is not None and len(local[lr % len(local)] & orig_affinity) >= 2:
                wide_affinity = local[lr % len(local)] & orig_affinity
            if pin_core is None:
                base = sorted(orig_affinity)
                usable = base[1:] if len(base) > 1 and base[0] == 0 else base
                pin_core = usable[lr % len(usable)]      # (one core per replica; no topology to go by)
                pin_source = "affinity mask (no GPU topology in sysfs)"
            os.sched_setaffinity(0, {pin_core})
        except Exception:
            pin_core = None
            pin_source = "none"

    # The HIP library is loaded before torch so that libamdhip64 resolves to /opt/rocm's copy.
    import liodom_amd as la
    from liodom_amd import synth
    from liodom_amd.replicas import Replicas
    la.load()
    rep = Replicas()          # one process per GPU; gloo rendezvous only when WORLD_SIZE > 1
    rank, local_rank, world = rep.rank, rep.local_rank, rep.world
    if args.gpus != world:
        # `--gpus N` without the N-rank launcher would report N x the throughput of one process
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d: launch with `python -m torch.distributed.run "
                             "--nnodes=1 --nproc-per-node %d ... bench.py --gpus %d`\n" % (args.gpus, world, args.gpus, args.gpus))
        rep.close()
        sys.exit(2)
    ndev = la.device_count()
    if ndev > 0 and world > ndev:            # more ranks than GPUs (smoke runs): share devices
        local_rank = local_rank % ndev
        # processes that share a GPU cannot see each other's handles: the in-kernel waits of the overlapped second kNN pass
        # are meant for a GPU one handle has to itself (the library switches it off for a second handle in ONE process)
        os.environ.setdefault("LIODOM_KNN_OVERLAP", "0")
    if pin_core is None:
        rep.pin_cpus(local_rank)             # host thread near the GPU's NUMA node (busy-polls the result record)

    wl = WORKLOADS[args.workload]
    H, W, R, epr, P = wl["H"], wl["W"], wl["R"], wl["epr"], wl["P"]
    N = H * W
    K, Wm = args.steps, args.warmup
    F = P                                    # untimed pre-fill: the window holds P frames before the warm-up starts
    total = F + Wm + K
    n_res = total + 1                        # (+ the scan whose extraction the last timed step issues)

    # ---- synthetic stream (stream id = global rank), generated before anything is timed ----
    cfg = synth.make_cfg(H, W, wl["lidar_type"])
    scans = [synth.scan(cfg, rep.stream_id, k)[0] for k in range(n_res)]
    if wl.get("ragged"):
        scans = [synth.ragged(x, H, W, wl["lidar_type"], seed=1000 * rep.stream_id + k) for k, x in enumerate(scans)]

    params = la.make_params(lidar_type=wl["lidar_type"], scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P)
    g = la.Liodom(params, la.make_config(device=local_rank, n_streams=1, max_points=N, max_width=W, pose_log_capacity=total + 8))
    g.alloc_resident(n_res)
    for k in range(n_res):
        g.upload_scan(0, k, scans[k])
    g.sync()
    time.sleep(0.5)   # let the host settle after the OpenMP-heavy generation (container CPU quota)

    def run(first, count, readback=True, pipelined=True, depth=1, ahead=False):
        # pipelined: the extraction of scan k+1 is issued on a second HIP stream while scan k's
        # odometry runs (the reference's own two-thread pipeline); never across the region's ends
        last = first + count - 1
        if readback and pipelined:
            # the consumer loop in C (liodom_replay_resident): every pose is read back, in order.  depth 1 (headline):
            # the odometry of scan k+1 is submitted before pose k is waited for — the device needs nothing from the host
            # between two scans, and the poses arrive exactly when they would anyway.  depth 0 (strict_sync leg): pose k
            # is read back before scan k+1's odometry is submitted; the GPU then idles for the host's turn-around.
            # ahead: the call also issues the extraction of the scan behind its last one, as every step inside it does for its
            # successor — the next call starts with that extraction already running beside the previous odometry (steady state)
            g.replay_resident(first, count, N, H, W, depth=depth, ahead=ahead)
            return
        for k in range(first, first + count):
            g.process_resident(k, N, H, W, readback=readback, next_slot=(k + 1 if (pipelined and k < last) else -1))

    # ---- timed region: untimed pre-fill + W warm-up steps, then exactly K steps, bracketed by sync + barrier.  A K-step region
    # lasts 1.6 ms at K = 20: one sample of it sits 6-7 % below the K = 200 figure on a freshly started process (clocks, first
    # launches of every kernel).  So the SAME region (reset, pre-fill, W warm-up steps, K timed steps: identical work, identical
    # poses) is measured `repeats` + 1 times; the first is discarded, `value` is the median, the spread is reported beside it. ----
    samples = []
    for r in range(max(1, args.repeats) + 1):
        if r:
            g.reset()
        # A step of the pipelined replay = the odometry of scan k + the extraction of scan k + 1 beside it.  The timed region
        # is K such steps in the steady state: the warm-up's last step has issued the extraction of the first timed scan (as
        # every step does for its successor), and the last timed step issues the extraction of the scan behind the region —
        # K odometries and K extractions inside the region, all completed before the clock stops (g.sync()).
        run(0, F)
        run(F, Wm, ahead=True)
        g.sync()
        rep.barrier()
        t0 = time.perf_counter()
        run(F + Wm, K, ahead=True)
        g.sync()
        rep.barrier()
        samples.append(rep.max_over_ranks(time.perf_counter() - t0))
    kept = sorted(samples[1:]) if len(samples) > 1 else samples
    elapsed = kept[len(kept) // 2] if len(kept) % 2 else 0.5 * (kept[len(kept) // 2 - 1] + kept[len(kept) // 2])
    poses_gpu, infos = g.pose_log(0, 0, total)
    status_bits = 0
    for i in infos:
        status_bits |= int(i.status)
    if status_bits:
        raise SystemExit("bench.py: the device raised status bits 0x%x (ring / edge / hash overflow): results invalid" % status_bits)
    value = world * K / elapsed
    timed = infos[F + Wm:]

    # ---- roofline leg: same K steps again with HIP events around every kernel launch ----
    g.reset()
    run(0, F + Wm)
    g.sync()
    g.reset_kernel_stats()
    g.set_profiling(True)
    run(F + Wm, K)
    stats = g.kernel_stats()
    g.set_profiling(False)
    meanE = float(np.mean([i.n_edges for i in timed]))
    meanM = float(np.mean([i.map_points for i in timed]))
    meanC = float(np.mean([(i.matches[0] + i.matches[1]) / 2.0 for i in timed]))
    mean_evals = float(np.mean([(i.lm[0].iterations + i.lm[1].iterations + 2) / 2.0 for i in timed]))
    roofline = roofline_from_stats(stats, 1, N, meanE, meanM, meanC, mean_evals, args.workload)
    roofline8d = roofline_8d(stats, 1, N, meanE, meanM, meanC, mean_evals)
    # Reference legs (never `value`): a K-step region lasts 1.5-3 ms at the driver's K = 20, and one sample of it swings by a
    # third from run to run (3 500-6 200 scans/s for the serial leg on one box) — the median of three, as for the drop-in legs.
    def ref_leg(**kw):
        t_ = []
        for _ in range(3):
            g.reset()
            run(0, F + Wm, **kw)
            g.sync()
            t1 = time.perf_counter()
            run(F + Wm, K, **kw)
            g.sync()
            t_.append(time.perf_counter() - t1)
        return K / sorted(t_)[1]
    async_rate = ref_leg(readback=False)          # asynchronous replay (no per-scan readback)
    strict_rate = ref_leg(depth=0)                # strictly synchronous consumer (pose k read back before scan k+1's odometry is submitted)
    serial_rate = ref_leg(pipelined=False)        # strictly serial scans (no overlap between extraction and odometry)
    modes = g.modes()
    # ---- drop-in-shaped legs (rank 0 of a 1-GPU run only; never `value`) ----
    host_fed = two_thread = None
    if world == 1:
        # (a) scans arrive in HOST memory: pinned ring, upload + extraction of scan k+1 overlap the odometry of scan k, every pose read back
        host = np.zeros((total, 1, N, 4), dtype=np.float32)
        for k in range(total):
            host[k, 0, :scans[k].shape[0]] = scans[k]
        # The host ring is page-locked ONCE, outside every timed region (a node assembles its clouds in a pinned ring that lives as long
        # as the node: liodom_scan_buffer / liodom_pin_host_buffer); earlier rounds registered and unregistered the K scans inside
        # the timed call.  Two figures: the single call over the K timed scans (cold start: the first scan's upload + extraction
        # overlap nothing, the pipeline drains at the end) and the STEADY-STATE rate = K / (T(2K scans) - T(K scans)), both replays
        # from the same pre-filled state — the difference removes those fixed costs, which at the driver's K = 20 are a sixth of the call.
        pinned = g.L.liodom_pin_host_buffer(host.ctypes.data_as(ctypes.c_void_p), host.nbytes) == 0
        # (2K scans: the K timed scans, then the same K scans walked backwards — a continuous trajectory, page-locked as one array)
        hf_src = np.ascontiguousarray(np.concatenate([host[F + Wm:F + Wm + K], host[F + Wm:F + Wm + K][::-1]]))
        src_pinned = g.L.liodom_pin_host_buffer(hf_src.ctypes.data_as(ctypes.c_void_p), hf_src.nbytes) == 0

        def hf_call(n_scans):
            g.reset()
            g.replay_host(host[:F + Wm], N, H, W, depth=1, pin=not pinned)
            t = time.perf_counter()
            p_, _ = g.replay_host(hf_src[:n_scans], N, H, W, depth=1, pin=not src_pinned)
            return time.perf_counter() - t, p_
        try:
            t_k, t_2k, hp = [], [], None
            for _ in range(3):
                a_, p_ = hf_call(K)
                t_k.append(a_)
                hp = p_ if hp is None else hp
                b_, _p2 = hf_call(2 * K)
                t_2k.append(b_)
        finally:
            if src_pinned:
                g.L.liodom_unpin_host_buffer(hf_src.ctypes.data_as(ctypes.c_void_p))
            if pinned:
                g.L.liodom_unpin_host_buffer(host.ctypes.data_as(ctypes.c_void_p))
        th = sorted(t_k)[1]
        th2 = sorted(t_2k)[1]
        # (continues the same trajectory: must equal the resident replay's poses)
        hf_ok = bool(np.array_equal(hp[:, 0].view(np.uint64), poses_gpu[F + Wm:F + Wm + K].view(np.uint64)))
        # (scans_per_s keeps its meaning of rounds 1-4 — one call over the K timed scans —; the differential figure has a name of its own
        #  and is only quoted when the difference of the two medians is a measurement: at least a quarter of the single call)
        steady = K / (th2 - th) if (th2 - th) > 0.25 * th else None
        host_fed = {"scans_per_s": round(K / th, 2), "us_per_scan": round(th / K * 1e6, 2),
                    "steady_state_scans_per_s": round(steady, 2) if steady else None,
                    "single_call_spread_s": [round(x, 6) for x in sorted(t_k)], "double_call_spread_s": [round(x, 6) for x in sorted(t_2k)],
                    "mode": "liodom_replay_host, depth 1: page-locked host ring (pinned once, outside the timed region) -> hipMemcpyAsync (%.2f MB per scan), "
                            "upload + extraction of scan k+1 beside the odometry of scan k, every pose read back in order; scans_per_s = one call over the "
                            "K timed scans, cold start and drain included (median of 3); steady_state_scans_per_s = K / (T(2K scans) - T(K scans)), "
                            "the second K scans being the first K walked backwards" % (N * 16 / 1e6),
                    "poses_bit_equal_to_resident_replay": hf_ok}
        # (b) two threads through the C-ABI, as the reference node runs its FeatureExtractor / LaserOdometer threads (liodom_node.cc:89-91):
        # two C++ threads (liodom_host_two_thread_replay, liodom_amd/host) — the extractor thread uploads every scan from host
        # memory, extracts and fetches the ~edges cloud, the edge cloud itself stays on the device (ticket queue), the odometer
        # thread reads every pose back
        g.reset()
        if pin_core is not None:                 # two busy threads from here on: the full mask again
            try:
                os.sched_setaffinity(0, wide_affinity)      # (the GPU's own socket where the topology is known)
            except Exception:
                pass
        nrun = F + Wm + K
        tt_all = []
        for _ in range(3):                         # (median of three, like every other leg)
            if tt_all:
                g.reset()
            tt_all.append(g.two_thread_replay(host[:nrun, 0], N, H, W, timed_from=F + Wm, fetch_edges=True, depth=1))
        tt_poses, tt, tt_edges = sorted(tt_all, key=lambda x: x[1])[1]
        two_thread = {"scans_per_s": round(K / tt, 2), "us_per_scan": round(tt / K * 1e6, 2),
                      "mode": "two C++ threads on one handle (liodom_host_two_thread_replay): extractor = liodom_extract_edges_device "
                              "(scan uploaded from page-locked host memory, %.2f MB) + liodom_wait_edges (the ~edges cloud, host copy); "
                              "ticket queue; odometer = liodom_odometry_submit_device / liodom_odometry_collect (depth 1, every pose read "
                              "back in order)" % (N * 16 / 1e6),
                      "edges_fetched": int(tt_edges),
                      "poses_bit_equal_to_resident_replay": bool(np.array_equal(tt_poses.view(np.uint64), poses_gpu[:nrun].view(np.uint64)))}
        g.reset()
        tt0_poses, tt0, _ = g.two_thread_replay(host[:nrun, 0], N, H, W, timed_from=F + Wm, fetch_edges=True, depth=0)
        two_thread["strict_sync_scans_per_s"] = round(K / tt0, 2)
        two_thread["poses_bit_equal_to_resident_replay"] = bool(two_thread["poses_bit_equal_to_resident_replay"] and
                                                                np.array_equal(tt0_poses.view(np.uint64), poses_gpu[:nrun].view(np.uint64)))
    dev_name, cus = g.device_info()
    g.close()

    out = None
    if rank == 0:
        out = {
            "metric": "scans/sec (64×1800 cloud, prev_frames=20) at 1 GPU; pose RMSE vs CPU ref",
            "value": round(value, 2), "unit": "scans/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(elapsed / K * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["name"], "streams_per_gpu": 1, "points_per_scan": N,
                       "mode": "pose of every scan read back in order by the consumer loop (liodom_replay_resident, depth 1: the "
                               "odometry of scan k+1 is submitted before pose k is waited for), scans resident in HBM, "
                               "extraction of scan k+1 overlapped with odometry of scan k on a second HIP stream; the timed region "
                               "is K steady-state steps = K odometries + K extractions (the step before it has issued the first "
                               "timed scan's extraction, the last timed step issues the next scan's)",
                       "prefill_scans": F,
                       "timed_region_repeats": len(kept), "timed_region_discarded": len(samples) - len(kept),
                       "value_is": "median over the repeats of the K-step timed region (same scans, same poses every repeat)",
                       "parallelism": "replicas only" if world > 1 else "single stream",
                       "mean_edges": round(meanE, 1), "mean_map_points": round(meanM, 1), "mean_matches": round(meanC, 1),
                       "mean_lm_evals_per_solve": round(mean_evals, 2), "device": dev_name, "compute_units": cus,
                       "library_source_hash": la.api.build_info().get("source_hash") or la.api.built_hash(),
                       "modes": modes,
                       "host_affinity": ("single-threaded legs (value, strict, async, serial, host_fed, batched): this process and the HIP runtime's threads on CPU %d "
                                         "(as `taskset -c`; LIODOM_BENCH_PIN=0: the CPU list of the GPU's NUMA node); "
                                         "(chosen from: %s); two_thread: %d CPUs of the GPU's socket; CPU baselines: the full mask" % (pin_core, pin_source, len(wide_affinity))) if pin_core is not None else "the GPU's NUMA node (no single core)",
                       "environment": {k: v for k, v in os.environ.items() if k.startswith("LIODOM_") or k in ("HIP_FORCE_DEV_KERNARG", "AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING")}},
            "value_spread": {"min": round(world * K / max(kept), 2), "max": round(world * K / min(kept), 2),
                             "first_discarded": round(world * K / samples[0], 2) if len(samples) > 1 else None},
            "strict_sync_scans_per_s": round(strict_rate, 2),
            "async_replay_scans_per_s": round(async_rate, 2),
            "serial_scans_per_s": round(serial_rate, 2),
            "roofline": roofline,
            "roofline_8d": roofline8d,
        }
        if host_fed:
            out["host_fed_scans_per_s"] = host_fed["scans_per_s"]          # PCIe-inclusive rate (never `value`)
            out["two_thread_scans_per_s"] = two_thread["scans_per_s"]      # the C-ABI as a patched liodom_node drives it
            out["host_fed"] = host_fed
            out["two_thread"] = two_thread
        if args.workload not in ("hdl64",):
            out["metric"] = "scans/sec (%dx%d cloud, prev_frames=%d) at 1 GPU; pose RMSE vs CPU ref" % (H, W, P)

    # ---- parity, every replica on its own stream (the oracle on the host cores of that rank) ----
    from oracle import oracle as orc
    n_par = args.parity_scans if args.parity_scans > 0 else F + Wm + min(K, 100)
    n_par = min(n_par, total)
    want_baseline = rank == 0 and world == 1 and not args.no_cpu_baseline
    n_cpu = min(K, max(1, args.cpu_sample))
    n_run = max(n_par, F + Wm + n_cpu) if want_baseline else n_par
    poses_cpu, tc = run_oracle(orc, wl, scans[:n_run], (1, 1), time_from=F + Wm)
    dt, dr = pose_errors(poses_gpu[:n_par], poses_cpu[:n_par])
    worst_t = rep.max_over_ranks(float(dt.max()))
    worst_r = rep.max_over_ranks(float(dr.max()))
    if rank == 0:
        out["parity"] = {
            "pose_trans_rmse_m": float(np.sqrt(np.mean(dt ** 2))), "pose_trans_max_m": float(dt.max()),
            "pose_rot_rmse_rad": float(np.sqrt(np.mean(dr ** 2))), "pose_rot_max_rad": float(dr.max()),
            "worst_replica_trans_max_m": worst_t, "worst_replica_rot_max_rad": worst_r, "replicas_checked": world,
            "tolerance": "1e-4 m / 1e-4 rad per scan", "scans_compared": int(n_par),
            "pass": bool(worst_t <= 1e-4 and worst_r <= 1e-4),
        }

    # ---- CPU baseline (rank 0, N = 1 only): bounded sample, 1 thread and the reference's thread policy ----
    if want_baseline:
        n_timed = n_run - (F + Wm)
        nproc = os.cpu_count() or 1
        st_threads, ev_threads = max(2, nproc - 5), nproc      # feature_extractor.cc:29-34, laser_odometry.cc:216
        # the reference's thread counts come from the machine, not from this process's affinity mask: give the OpenMP
        # threads the CPUs back that pin_cpus() took away for the GPU legs (nproc threads on one NUMA node's cores
        # would time an oversubscribed run)
        try:
            os.sched_setaffinity(0, orig_affinity)
        except Exception:
            pass
        _, tc_ref = run_oracle(orc, wl, scans[:n_run], (st_threads, ev_threads), time_from=F + Wm)
        try:
            cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            cpu_model = "unknown"
        out["cpu_baseline"] = {
            "value": round(n_timed / tc, 3), "unit": "scans/s", "cores": 1, "kind": "port",
            "sample": "scans %d..%d of the same stream (%d timed after %d untimed pre-fill + warm-up scans), CPU oracle "
                      "(oracle/liodom_oracle.cc, kd-tree kNN rebuilt twice per scan), 1 thread of %d" % (F + Wm, n_run - 1, n_timed, F + Wm, nproc),
            "gpu_over_cpu": round(value / (n_timed / tc), 1),
            "reference_policy": {
                "value": round(n_timed / tc_ref, 3), "unit": "scans/s", "cores": nproc,
                "stencil_threads": st_threads, "residual_eval_threads": ev_threads,
                "policy": "stencil loop omp num_threads = max(2, nproc - 5) (feature_extractor.cc:29-34,194); residual blocks "
                          "evaluated by nproc threads (laser_odometry.cc:216); everything else serial as in the reference",
                "OMP_WAIT_POLICY": os.environ.get("OMP_WAIT_POLICY"), "OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"),
            },
            "nproc": nproc, "cpu_model": cpu_model,
        }

    # ---- batched leg: lock-step streams on one GPU (throughput mode; separately labelled) ----
    if rank == 0 and args.batched_streams > 0 and world == 1:
        S = args.batched_streams
        if pin_core is not None:
            try:
                os.sched_setaffinity(0, {pin_core})      # (single enqueueing thread again)
            except Exception:
                pass
        Kb, Wb = min(K, 20), F + min(Wm, 4)      # pre-fill + a few warm-up steps, then Kb timed steps
        tb = Kb + Wb
        n_data = max(1, min(args.batched_data_streams, S))   # distinct synthetic streams; stream s replays data stream s % n_data
        data = [scans[:tb]] + [[synth.scan(cfg, 1000 + d, k)[0] for k in range(tb)] for d in range(1, n_data)]
        if wl.get("ragged"):
            data = [data[0]] + [[synth.ragged(x, H, W, wl["lidar_type"], seed=1000 * (1000 + d) + k) for k, x in enumerate(data[d])] for d in range(1, n_data)]
        gb = la.Liodom(params, la.make_config(device=local_rank, n_streams=S, max_points=N, max_width=W, pose_log_capacity=tb + 8))
        bmodes = gb.modes()
        gb.alloc_resident(tb)
        for s in range(S):
            for k in range(tb):
                gb.upload_scan(s, k, data[s % n_data][k])
        gb.sync()
        time.sleep(0.5)
        for k in range(Wb):
            gb.process_resident(k, N, H, W, readback=True, next_slot=(k + 1 if k + 1 < Wb else -1))
        gb.sync()
        t2 = time.perf_counter()
        for k in range(Wb, tb):      # same mode as the headline: per-step synchronous, next extraction overlapped
            gb.process_resident(k, N, H, W, readback=True, next_slot=(k + 1 if k + 1 < tb else -1))
        gb.sync()
        eb = time.perf_counter() - t2
        bposes, binfos = gb.pose_log(0, 0, tb)
        bstatus = 0
        for s in range(0, S, max(1, S // 8)):
            for i in gb.pose_log(s, 0, tb)[1]:
                bstatus |= int(i.status)
        # lock-step stream 0 replays the headline data stream: same poses as the single-stream handle
        bdt, bdr = pose_errors(bposes[:min(tb, n_par)], poses_gpu[:min(tb, n_par)])
        gb.reset()
        for k in range(Wb):
            gb.process_resident(k, N, H, W, readback=True)
        gb.reset_kernel_stats()
        gb.set_profiling(True)
        for k in range(Wb, tb):
            gb.process_resident(k, N, H, W, readback=True)
        bstats = gb.kernel_stats()
        gb.set_profiling(False)
        gb.close()
        bt = binfos[Wb:]
        bE = float(np.mean([i.n_edges for i in bt]))
        bM = float(np.mean([i.map_points for i in bt]))
        bC = float(np.mean([(i.matches[0] + i.matches[1]) / 2.0 for i in bt]))
        bev = float(np.mean([(i.lm[0].iterations + i.lm[1].iterations + 2) / 2.0 for i in bt]))
        out["batched"] = {
            "streams": S, "steps": Kb, "warmup": Wb, "value": round(S * Kb / eb, 1), "unit": "scans/s (aggregate, 1 GPU)",
            "ms_per_step": round(eb / Kb * 1e3, 4), "distinct_data_streams": n_data, "status_bits": bstatus,
            "stream0_vs_single_stream_max_m": float(bdt.max()), "stream0_vs_single_stream_max_rad": float(bdr.max()),
            "note": "lock-step streams in one launch per kernel, per-step synchronous, extraction of step k+1 overlapped; "
                    "%d distinct synthetic streams replayed round-robin" % n_data,
            "roofline": roofline_from_stats(bstats, S, N, bE, bM, bC, bev, args.workload),
            "roofline_8d": roofline_8d(bstats, S, N, bE, bM, bC, bev),
            "modes": bmodes,
        }

    rep.close()
    if rank == 0:
        print(json.dumps(out))
        if not out["parity"]["pass"] or (host_fed and not (host_fed["poses_bit_equal_to_resident_replay"] and two_thread["poses_bit_equal_to_resident_replay"])):
            sys.exit(3)


if __name__ == "__main__":
    main()

"""What the speculative hand-overs (kernels_sync.h) are worth OFF the benchmark stream: the headline shape (64 x 1800, P = 20) replayed in
chain mode (liodom_replay_resident, depth 1: the loop bench.py's `value` times) on workloads with other convergence statistics, with
LIODOM_SPECULATE = 0 and 1: scans/s, and per solve how many iterates left early / were not confirmed (liodom_get_modes: spec_early,
spec_unconfirmed; first solve / finalising solve).  Workloads: base = bench.py's stream; ragged = synth.ragged (~25 % no-returns, unequal
rings); noise3cm = range noise sigma 3 cm instead of 1 cm; yawstep2 / yawstep5 = an abrupt extra yaw of 2 / 5 degrees every 10 scans
(the cloud rotated about the sensor's z axis: rings and ranges unchanged).  The GPU poses are compared with the oracle's on the
first P + 30 scans of every workload (5 degrees loses track in the reference's own algorithm: both sides must lose it alike).
usage: python tools/spec_workloads.py [timed scans = 200] [repeats = 5]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("SPEC_PIN", "1") != "0":
    try:
        from liodom_amd.replicas import gpu_local_cpus, choose_core
        loc = gpu_local_cpus()
        core = choose_core(0, loc, os.sched_getaffinity(0)) if loc else None
        if core is not None:
            os.sched_setaffinity(0, {core})
    except Exception:
        pass
import liodom_amd as la
from liodom_amd import synth
from oracle import oracle as orc
K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H, W, R, epr, P = 64, 1800, 8, 10, 20
N = H * W
Wm = 20
total = P + Wm + K


def rotz(x, deg):
    c, s = np.cos(np.radians(deg)), np.sin(np.radians(deg))
    y = x.copy()
    y[:, 0] = c * x[:, 0] - s * x[:, 1]
    y[:, 1] = s * x[:, 0] + c * x[:, 1]
    return y.astype(np.float32)


def workload(name):
    cfg = synth.make_cfg(H, W, 0, noise_sigma=0.03 if name == "noise3cm" else 0.01)
    scans = [synth.scan(cfg, 0, k)[0] for k in range(total)]
    if name == "ragged":
        scans = [synth.ragged(x, H, W, 0, seed=k) for k, x in enumerate(scans)]
    if name.startswith("yawstep"):
        step = float(name[len("yawstep"):])
        scans = [rotz(x, -step * (k // 10)) for k, x in enumerate(scans)]
    return scans


print("# tools/spec_workloads.py %d %d: headline shape, chain mode, depth-1 resident replay, %d timed scans behind %d pre-fill + %d warm-up, median of %d" % (K, REP, K, P, Wm, REP))
print("%-10s %-10s %10s %10s   %-22s %-22s %s" % ("workload", "speculate", "scans/s", "us/scan", "first solve early/unconf", "finalising early/unconf", "max |dt| m, |dr| rad vs oracle (%d scans); LM iterations / solve" % (P + 30)))
for name in ("base", "ragged", "noise3cm", "yawstep2", "yawstep5"):
    scans = workload(name)
    po = orc.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P, knn_mode=1)
    od = orc.Odometer(po)
    ref = []
    for k in range(P + 30):
        e = orc.extract(po, scans[k], H, W)
        ref.append(od.step(e["edges"])[0].copy())
    od.close()
    ref = np.array(ref)
    res = {}
    for spec in ("0", "1"):
        os.environ["LIODOM_SPECULATE"] = spec
        g = la.Liodom(la.make_params(scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
                      la.make_config(n_streams=1, max_points=N, max_width=W, pose_log_capacity=total + 8))
        g.alloc_resident(total + 1)
        for k in range(total):
            g.upload_scan(0, k, scans[k])
        g.upload_scan(0, total, scans[total - 1])
        g.sync()
        rates = []
        for r in range(REP + 1):
            g.reset()
            g.replay_resident(0, P, N, H, W, depth=1)
            g.replay_resident(P, Wm, N, H, W, depth=1, ahead=True)
            g.sync()
            t0 = time.perf_counter()
            poses, infos = g.replay_resident(P + Wm, K, N, H, W, depth=1, ahead=True)
            g.sync()
            if r:
                rates.append(K / (time.perf_counter() - t0))
        m = g.modes()
        pl, il = g.pose_log(0, 0, total)
        st = 0
        for i in il:
            st |= int(i.status)
        dt = np.linalg.norm(pl[:P + 30, 4:] - ref[:, 4:], axis=1).max()
        dots = np.abs(np.sum(pl[:P + 30, :4] * ref[:, :4], axis=1))
        dr = (2.0 * np.arccos(np.minimum(1.0, dots))).max()
        its = np.mean([(i.lm[0].iterations + i.lm[1].iterations) / 2.0 for i in il[P + Wm:]])
        e0, e1 = m["spec_early"].split("/")
        u0, u1 = m["spec_unconfirmed"].split("/")
        rate = float(np.median(rates))
        res[spec] = rate
        # (counters cover one pass over pre-fill + warm-up + K scans: the last repeat since its reset)
        print("%-10s %-10s %10.0f %10.2f   %-22s %-22s %.1e, %.1e; %.2f%s" % (name, spec, rate, 1e6 / rate, "%s / %s of %d" % (e0, u0, total), "%s / %s of %d" % (e1, u1, total), dt, dr, its,
                                                                            "" if st == 0 else "  STATUS 0x%x" % st), flush=True)
        assert m["chain"] == "1" and m["speculate"] == spec, m
        g.close()
    print("%-10s speculation on / off: %+.1f %%" % (name, 100.0 * (res["1"] / res["0"] - 1.0)), flush=True)

"""Speculative hand-over across mode switches: a replay that enters and leaves chain mode (per-kernel profiling on / off, plain
entry points, synchronisations and getters in between) with the predictor forced wrong (LIODOM_SPECULATE=2: every hand-over is
repaired — by k_chain_redo0 behind the next scan's first pass, or, where no such pass follows, by the repair the host enqueues on its
own) must give the pose log of a straight replay without speculation, bit for bit.
usage: python tools/spec_switches.py [hdl64|vlp16|16x900] [scans]   (spawns itself once per mode: the switch is read at handle creation)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shape = sys.argv[1] if len(sys.argv) > 1 else "hdl64"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 240
if len(sys.argv) > 3:
    import liodom_amd as la
    from liodom_amd import synth
    H, W, LT, R, epr, P = {"hdl64": (64, 1800, 0, 8, 10, 20), "vlp16": (16, 1800, 0, 8, 20, 10), "16x900": (16, 900, 0, 6, 10, 5)}[shape]
    N = H * W
    cfg = synth.make_cfg(H, W, LT)
    g = la.Liodom(la.make_params(lidar_type=LT, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
                  la.make_config(n_streams=1, max_points=N, max_width=W, pose_log_capacity=K + 8))
    g.alloc_resident(K)
    for k in range(K):
        g.upload_scan(0, k, synth.scan(cfg, 7, k)[0])
    g.sync()
    straight = sys.argv[4] == "straight"
    out = []
    if straight:
        p, _ = g.replay_resident(0, K, N, H, W, depth=1)
        out.append(p[:, 0].copy())
    else:
        k = 0
        seg = 0
        while k < K:
            n = min(17 + 5 * (seg % 3), K - k)
            kind = seg % 5
            if kind == 1:
                g.set_profiling(True)              # per-kernel profiling: no overlap, no chain
            if kind == 3:                          # per-call pipelined path, then a plain call (extraction on the odometry stream)
                for i in range(n):
                    nxt = k + i + 1 if i + 1 < n else -1
                    pose, info = g.process_resident(k + i, N, H, W, readback=True, next_slot=nxt)
                    assert int(info[0].status) == 0
                    out.append(pose.reshape(1, 7).copy())
            else:
                p, infos = g.replay_resident(k, n, N, H, W, depth=seg % 2)
                assert all(int(i.status) == 0 for i in infos)
                out.append(p[:, 0].copy())
            if kind == 1:
                g.set_profiling(False)
            if kind == 2:
                g.sync()
                g.pose_log(0, 0, k + n)            # a getter that reads the odometry side's device results
            k += n
            seg += 1
    np.save(sys.argv[3], np.concatenate(out))
    m = g.modes()
    print("chain %s speculate %s" % (m.get("chain"), m.get("speculate")), flush=True)
    g.close()
    sys.exit(0)
files = []
for tag, spec, how in (("ref", "0", "straight"), ("forced", "2", "switches"), ("model", "1", "switches")):
    f = "/tmp/specsw_%s_%s.npy" % (shape, tag)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), shape, str(K), f, how], env=dict(os.environ, LIODOM_SPECULATE=spec), capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    files.append(f)
ref = np.load(files[0])
bad = []
for f, nm in zip(files[1:], ("predictor forced wrong", "model predictor")):
    x = np.load(f)
    d = np.nonzero(np.any(x.view(np.uint64) != ref.view(np.uint64), axis=1))[0]
    if len(d):
        bad.append("%s: %d scans differ, first %s" % (nm, len(d), d[:5]))
print("%s: %d scans, mode switches with the speculative hand-over (forced wrong / model) vs straight replay without: %s" % (shape, K, "bit-identical" if not bad else "DIFFERENT " + "; ".join(bad)))
sys.exit(0 if not bad else 1)

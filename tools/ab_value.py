"""Interleaved A/B of the headline rate between library variants (experiments; the product library is `product`).
usage: python tools/ab_value.py [--rounds 4] [--workload hdl64] [--scans 240] name[:ENV=VAL,...] ...
  name = `product` or a variant built by tools/variant_build.sh; optional environment per arm after a colon.
Every arm runs in its own process per round (A B C A B C ...): window pre-fill, then `scans` timed scans of the depth-1 resident
replay, 5 repeats, median; prints per arm the median over the rounds, the spread and the SHA-1 of the pose log (arms that
must not change results must agree on it)."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WL = {"hdl64": (64, 1800, 0, 8, 10, 20), "vlp16": (16, 1800, 0, 8, 20, 10), "ouster128": (128, 2048, 1, 8, 10, 30)}

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    name, wl, K = sys.argv[2], sys.argv[3], int(sys.argv[4])
    import liodom_amd.api as api
    if name != "product":
        api._LIB = os.path.join(ROOT, "build", "variants", "lib%s.so" % name)
        api.is_stale = lambda: False
    import time
    import numpy as np
    import liodom_amd as la
    from liodom_amd import synth
    H, W, lt, R, epr, P = WL[wl]
    cfg = synth.make_cfg(H, W, lt)
    F = P + 10
    scans = [synth.scan(cfg, 0, k)[0] for k in range(F + K + 1)]
    g = la.Liodom(la.make_params(lidar_type=lt, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
                  la.make_config(n_streams=1, max_points=H * W, max_width=W, pose_log_capacity=F + K + 8))
    g.alloc_resident(F + K + 1)
    for k, x in enumerate(scans):
        g.upload_scan(0, k, x)
    g.sync()
    rates = []
    for r in range(6):
        g.reset()
        g.replay_resident(0, F, H * W, H, W, depth=1, ahead=True)
        g.sync()
        t0 = time.perf_counter()
        g.replay_resident(F, K, H * W, H, W, depth=1, ahead=True)
        g.sync()
        rates.append(K / (time.perf_counter() - t0))
    poses, infos = g.pose_log(0, 0, F + K)
    st = 0
    for i in infos:
        st |= int(i.status)
    rates = sorted(rates[1:])
    print("RESULT %.1f %s %d" % (rates[len(rates) // 2], hashlib.sha1(poses.tobytes()).hexdigest()[:12], st), flush=True)
    g.close()
    sys.exit(0)

args = sys.argv[1:]
rounds, wl, K = 4, "hdl64", 240
while args and args[0].startswith("--"):
    if args[0] == "--rounds": rounds = int(args[1])
    elif args[0] == "--workload": wl = args[1]
    elif args[0] == "--scans": K = int(args[1])
    args = args[2:]
arms = []
for a in args:
    name, _, envs = a.partition(":")
    env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
    arms.append((a, name, env))
res = {a: [] for a, _, _ in arms}
sha = {}
for r in range(rounds):
    for a, name, env in arms:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, wl, str(K)], env=dict(os.environ, **env),
                           capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        if not line:
            print(a, "FAILED:", p.stderr[-1500:]); continue
        _, rate, h, st = line[-1].split()
        res[a].append(float(rate)); sha.setdefault(a, set()).add(h + ("!status=%s" % st if st != "0" else ""))
base = None
for a, _, _ in arms:
    v = sorted(res[a])
    if not v: continue
    med = v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
    if base is None: base = med
    print("%-40s median %8.1f scans/s (%+.2f %%)  min %8.1f max %8.1f  poses %s" % (a, med, 100.0 * (med / base - 1.0), v[0], v[-1], ",".join(sorted(sha[a]))))

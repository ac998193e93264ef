"""Soak of the in-kernel cross-stream signalling (pipe flags, overlapped second kNN pass, in-launch exchanges) while a SECOND
PROCESS saturates the same GPU.  The library can only see the handles of its own process (g_live_handles); nothing tells it
that another process's kernels occupy the CUs its waiting workgroups expect to share.  Outcome required: the replay's pose log
is bit-identical to the solo run — or the library reports a clean LIODOM_ERR_HIP (LIODOM_STATUS_PIPE_TIMEOUT), and after
liodom_reset() (event-based dependencies from then on) the replay completes bit-identical.  Never a wrong pose.

usage: python tools/soak_two_process.py [hdl64|vlp16] [scans=20000] [distinct=400] [hog=matmul|spin]
The replay walks `distinct` synthetic scans back and forth (0 .. distinct-1, distinct-2 .. 0, ...): a continuous trajectory of
any length from a bounded number of generated scans."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HOG = os.path.join(ROOT, "build", "gpu_hog")          # tools/gpu_hog.hip, built by __graft_entry__.build() (no torch: its first import on a fresh box takes minutes)
if not os.path.exists(HOG):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", HOG, os.path.join(ROOT, "tools", "gpu_hog.hip")])

import liodom_amd as la
from liodom_amd import synth
shape = sys.argv[1] if len(sys.argv) > 1 else "hdl64"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
D = int(sys.argv[3]) if len(sys.argv) > 3 else 400
hog_kind = sys.argv[4] if len(sys.argv) > 4 else "matmul"
H, W, LT, R, epr, P = {"hdl64": (64, 1800, 0, 8, 10, 20), "vlp16": (16, 1800, 0, 8, 20, 10)}[shape]
N = H * W
cfg = synth.make_cfg(H, W, LT)
print("generating %d scans ..." % D, flush=True)
scans = [synth.scan(cfg, 5, k)[0] for k in range(D)]
order = list(range(D)) + list(range(D - 2, 0, -1))            # one period of the back-and-forth walk
def make():
    h = la.Liodom(la.make_params(lidar_type=LT, scan_lines=H, scan_regions=R, edges_per_region=epr, prev_frames=P),
                  la.make_config(n_streams=1, max_points=N, max_width=W, pose_log_capacity=1024))
    h.alloc_resident(D)
    for k in range(D):
        h.upload_scan(0, k, scans[k])
    h.sync()
    return h


BUDGET_S = float(os.environ.get("SOAK_BUDGET_S", "25"))


def replay(g, label, budget=None):
    """K scans along the walk — or, with a time budget, as many of them as fit into it (at least one run of the walk): beside a
    second process the GPU may be time-sliced between the two, and the replay then advances at tens of milliseconds per scan
    whatever the library does; the property under test, never a wrong pose, is checked on the scans that ran.
    Returns (poses or None, error text or None)."""
    out = []
    t0 = time.perf_counter()
    k = 0
    try:
        while k < K:
            if budget is not None and k > 0 and time.perf_counter() - t0 > budget:
                print("%s: time budget of %.0f s used after %d of %d scans" % (label, budget, k, K), flush=True)
                break
            # runs of consecutive resident slots (ascending) go through the C consumer loop; descending stretches scan by scan
            i = k % len(order)
            if i < D - 1:
                n = min(D - i, K - k)
                if budget is not None:
                    n = min(n, 40)          # (beside the second process a scan can take tens of milliseconds: look at the clock often)
                poses, infos = g.replay_resident(order[i], n, N, H, W, depth=1)
                st = 0
                for inf in infos:
                    st |= int(inf.status)
                if st:
                    return None, "status bits 0x%x" % st
                out.append(poses[:, 0].copy())
                k += n
            else:
                nxt = order[(i + 1) % len(order)] if k + 1 < K else -1          # (its extraction is issued beside this odometry)
                pose, info = g.process_resident(order[i], N, H, W, readback=True, next_slot=nxt)
                if int(info[0].status):
                    return None, "status bits 0x%x" % int(info[0].status)
                out.append(pose.copy())
                k += 1
    except la.LiodomError as ex:
        return None, str(ex)
    dt = time.perf_counter() - t0
    print("%s: %d scans in %.2f s = %.0f scans/s" % (label, k, dt, k / dt), flush=True)
    return np.concatenate(out), None


# reference of the safe mode (what a handle falls back to after a timeout: one workgroup per solve, so its sums — and the last
# bits of its poses — differ from the normal mode's), from a handle created in it
os.environ["LIODOM_SAFE_MODE"] = "1"
gs = make()
solo_safe, err = replay(gs, "solo, safe mode %s" % gs.modes().get("safe_mode"))
assert err is None, err
gs.close()
del os.environ["LIODOM_SAFE_MODE"]
g = make()
print("modes:", g.modes(), flush=True)
solo, err = replay(g, "solo")
assert err is None, err
dq = np.abs(solo - solo_safe).max()
print("normal vs safe mode, solo: max |pose difference| %.3e" % dq, flush=True)
assert dq < 1e-6
stop = "/tmp/liodom_soak_stop_%d" % os.getpid()
for f in (stop, stop + ".ready"):
    if os.path.exists(f):
        os.remove(f)
def _die_with_parent():
    import ctypes
    ctypes.CDLL("libc.so.6").prctl(1, 9)      # PR_SET_PDEATHSIG = SIGKILL: the hog must not outlive a killed soak (it would keep the GPU busy — and inherited pipes open — for ever)


hog = subprocess.Popen([HOG, hog_kind, stop], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, preexec_fn=_die_with_parent)
t_wait = time.time()
while not os.path.exists(stop + ".ready") and time.time() - t_wait < 180:
    time.sleep(0.2)
assert os.path.exists(stop + ".ready"), "the second process did not start"
time.sleep(1.0)
verdict = "FAIL"
n_checked = 0
try:
    g.reset()
    loaded, err = replay(g, "beside the second process (%s)" % hog_kind, budget=BUDGET_S)
    if err is None:
        same = np.array_equal(loaded.view(np.uint64), solo[:len(loaded)].view(np.uint64))
        n_checked = len(loaded)
        verdict = "bit-identical to the solo run" if same else "WRONG POSES (no error reported)"
    else:
        print("clean error beside the second process:", err[:300], flush=True)
        g.reset()                                   # applies the event-path fallback
        print("modes after reset:", g.modes(), flush=True)
        again, err2 = replay(g, "after liodom_reset (safe mode), still beside the second process", budget=BUDGET_S)
        n_checked = 0 if again is None else len(again)
        if err2 is None and np.array_equal(again.view(np.uint64), solo_safe[:len(again)].view(np.uint64)):
            verdict = "clean LIODOM_ERR_HIP, then bit-identical to the solo safe-mode run after liodom_reset"
        else:
            verdict = "FAIL after the fallback: %s" % (err2 or "poses differ")
finally:
    open(stop, "w").write("1")
    try:
        print(hog.communicate(timeout=120)[0].strip())
    except Exception:
        hog.kill()
    for f in (stop, stop + ".ready"):
        if os.path.exists(f):
            os.remove(f)
print("%s, %d scans, overlap %s, %d poses compared beside the second process: %s" % (shape, K, g.modes().get("knn_overlap"), n_checked, verdict))
g.close()
sys.exit(0 if verdict.startswith(("bit-identical", "clean")) else 1)

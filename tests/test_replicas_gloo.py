"""N > 1 path of bench.py: one process per GPU, replicas only.  Covered here with two gloo
processes on CPU: rendezvous on 127.0.0.1, distinct stream per rank, barrier, max-over-ranks."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_replicas_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent("""
        import sys, time
        sys.path.insert(0, %r)
        from liodom_amd.replicas import Replicas
        rep = Replicas()
        assert rep.world == 2 and rep.stream_id == rep.rank and rep.device == rep.local_rank
        rep.barrier()
        elapsed = 1.0 + rep.rank            # rank 1 is "slower"
        mx = rep.max_over_ranks(elapsed)
        total = rep.sum_over_ranks(10 * (rep.rank + 1))
        assert mx == 2.0 and total == 30.0, (mx, total)
        # whole-job throughput = units of all ranks / max time
        open(%r + "/rank%%d.txt" %% rep.rank, "w").write("OK %%.1f" %% (2 * 100 / mx))
        rep.close()
    """ % (ROOT, str(tmp_path))))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    # (stdout of the two ranks may interleave, so each rank reports through its own file)
    assert (tmp_path / "rank0.txt").read_text() == "OK 100.0" and (tmp_path / "rank1.txt").read_text() == "OK 100.0"


def test_single_process_is_a_noop():
    sys.path.insert(0, ROOT)
    from liodom_amd.replicas import Replicas
    env_backup = {k: os.environ.pop(k, None) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    try:
        rep = Replicas()
        assert rep.world == 1 and rep.dist is None
        assert rep.max_over_ranks(3.5) == 3.5
        rep.barrier()
        rep.close()
    finally:
        for k, v in env_backup.items():
            if v is not None:
                os.environ[k] = v


def test_bench_refuses_gpus_without_the_launcher():
    """`bench.py --gpus N` must run under the N-rank launcher: one process reporting N x its own throughput would be a
    made-up aggregate.  The check comes before anything touches a GPU (exit code 2, nothing on stdout)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert out.returncode == 2, (out.returncode, out.stderr[-500:])
    assert out.stdout.strip() == "" and "WORLD_SIZE is 1" in out.stderr


def test_bench_under_the_launcher_agrees_on_the_world(tmp_path):
    """Two ranks of bench.py under torch.distributed.run on a box without GPUs: both pass the launcher check (--gpus ==
    WORLD_SIZE) and then fail loudly at the device — the product path has no CPU fall-back — with a non-zero exit."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("on a GPU box this is a real two-rank run: the driver's scaling bench covers it")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert out.returncode != 0
    assert "WORLD_SIZE is" not in out.stderr          # the launcher check passed on both ranks


def test_replica_cores_come_from_the_gpus_own_socket(tmp_path, monkeypatch):
    """bench.py pins every replica to ONE core before the HIP runtime loads (its threads inherit the mask).  The core must lie in the
    local_cpulist of the replica's own GPU, differ between replicas that share a socket, and avoid CPU 0 — round 5 took
    sorted(affinity)[LOCAL_RANK], i.e. socket 0 for all eight replicas of a two-socket node.  Mocked sysfs: 8 GPUs, 4 per socket."""
    sys.path.insert(0, ROOT)
    from liodom_amd.replicas import gpu_local_cpus, choose_core
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    root = tmp_path / "sys"
    nodes = root / "class" / "kfd" / "kfd" / "topology" / "nodes"
    # two CPU nodes without SIMDs, then eight GPUs
    for n in range(2):
        (nodes / str(n)).mkdir(parents=True)
        (nodes / str(n) / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for g in range(8):
        bus = 0x10 + 0x10 * g
        d = nodes / str(2 + g)
        d.mkdir(parents=True)
        d.joinpath("properties").write_text("cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain 0\n" % (bus << 8))
        pci = root / "bus" / "pci" / "devices" / ("0000:%02x:00.0" % bus)
        pci.mkdir(parents=True)
        pci.joinpath("local_cpulist").write_text("0-63,128-191\n" if g < 4 else "64-127,192-255\n")
    local = gpu_local_cpus(str(root))
    assert len(local) == 8 and 0 in local[0] and 64 in local[7] and 0 not in local[7]
    affinity = set(range(256))
    cores = [choose_core(g, local, affinity) for g in range(8)]
    assert all(cores[g] in local[g] for g in range(8)), cores
    assert len(set(cores)) == 8 and 0 not in cores, cores
    assert all(c < 64 or 128 <= c < 192 for c in cores[:4]) and all(64 <= c < 128 or c >= 192 for c in cores[4:]), cores
    # a restricted affinity mask (container quota) is respected; a GPU without a usable local core yields None
    assert choose_core(5, local, set(range(64, 72))) in range(64, 72)
    assert choose_core(5, local, set(range(0, 8))) is None
    # HIP_VISIBLE_DEVICES re-orders the devices the runtime will see
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "6,1")
    sel = gpu_local_cpus(str(root))
    assert len(sel) == 2 and 64 in sel[0] and 0 in sel[1]
    assert gpu_local_cpus(str(tmp_path / "nothing")) == []

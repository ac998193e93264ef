"""Replicas-only multi-GPU support.

The LiODOM hot path does not shard (every scan depends on the previous pose and window,
SURVEY.md §8e), so N GPUs run N independent replayed streams — one process per GPU, no collective
on the data path.  torch.distributed is used only for the rendezvous, the barrier around the
timed region and the max-over-ranks of the elapsed time.  The gloo backend is enough for that and
keeps torch's HIP runtime uninitialised (libliodom_hip owns the GPU in each process).
"""
import os


class Replicas:
    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29512")
            if not dist.is_initialized():
                # gloo prints its "[Gloo] Rank N is connected ..." banner on stdout from C++; the
                # bench contract is ONE JSON line on stdout, so route fd 1 to stderr meanwhile
                import sys
                sys.stdout.flush()
                saved = os.dup(1)
                try:
                    os.dup2(2, 1)
                    dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
                    dist.barrier()
                finally:
                    sys.stdout.flush()
                    os.dup2(saved, 1)
                    os.close(saved)
            self.dist = dist

    @property
    def stream_id(self):
        """Synthetic stream replayed by this replica (seeds are 1000*stream + scan)."""
        return self.rank

    @property
    def device(self):
        return self.local_rank

    def pin_cpus(self, device):
        """Best effort: restrict this process to the CPUs of the NUMA node the GPU hangs off
        (/sys/bus/pci/devices/<bus id>/local_cpulist) — the host thread busy-polls the host-mapped result
        record of every scan, and N replicas should not share one socket's cores by accident.  Returns
        the CPU set in use (None if nothing was changed)."""
        self.cpus = None
        try:
            from . import api
            bus = api.device_pci_bus_id(device)
            if not bus:
                return None
            path = "/sys/bus/pci/devices/%s/local_cpulist" % bus.lower()
            if not os.path.exists(path):
                return None
            cpus = set()
            for part in open(path).read().strip().split(","):
                if not part:
                    continue
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
            allowed = os.sched_getaffinity(0) & cpus
            if allowed:
                os.sched_setaffinity(0, allowed)
                self.cpus = sorted(allowed)
        except Exception:
            return None
        return self.cpus

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if not self.dist:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if not self.dist:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None

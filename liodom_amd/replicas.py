"""Replicas-only multi-GPU support.

The LiODOM hot path does not shard (every scan depends on the previous pose and window,
SURVEY.md §8e), so N GPUs run N independent replayed streams — one process per GPU, no collective
on the data path.  torch.distributed is used only for the rendezvous, the barrier around the
timed region and the max-over-ranks of the elapsed time.  The gloo backend is enough for that and
keeps torch's HIP runtime uninitialised (libliodom_hip owns the GPU in each process).
"""
import os


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(sysfs_root="/sys"):
    """CPU sets local to every GPU of this node, in HIP device order, read from sysfs WITHOUT loading the HIP runtime (its worker
    threads inherit the affinity mask of the thread that loads it, so the mask has to be chosen first): the KFD topology nodes with
    SIMDs are the GPUs, in the order the runtime enumerates them; `domain` / `location_id` (bus << 8 | device << 3 | function) give
    the PCI address, whose `local_cpulist` names the cores of the socket the GPU hangs off.  HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES
    (plain index lists) are honoured.  Returns a list of sets; [] if the topology cannot be read."""
    out = []
    try:
        base = os.path.join(sysfs_root, "class", "kfd", "kfd", "topology", "nodes")
        for n in sorted((d for d in os.listdir(base) if d.isdigit()), key=int):
            props = {}
            for line in open(os.path.join(base, n, "properties")):
                k, _, val = line.strip().partition(" ")
                props[k] = val
            if int(props.get("simd_count", "0")) <= 0:
                continue
            loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
            bdf = "%04x:%02x:%02x.%d" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7)
            path = os.path.join(sysfs_root, "bus", "pci", "devices", bdf, "local_cpulist")
            out.append(_parse_cpulist(open(path).read()) if os.path.exists(path) else set())
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
            sel = os.environ.get(var, "").strip()
            if sel and all(t.strip().isdigit() for t in sel.split(",")):
                out = [out[int(t)] for t in sel.split(",") if int(t) < len(out)]
    except Exception:
        return []
    return out


def choose_core(device, local_cpus, affinity):
    """One core for the process that drives GPU `device`: from the GPU's local CPU list (within the process's affinity), distinct for
    the replicas whose GPUs share a socket (the k-th GPU of a socket takes the socket's k-th usable core), never CPU 0 when there is
    a choice (IRQs and housekeeping land there).  None if nothing is known."""
    if device < 0 or device >= len(local_cpus):
        return None
    mine = sorted(local_cpus[device] & set(affinity))
    if len(mine) > 1 and mine[0] == 0:
        mine = mine[1:]
    if not mine:
        return None
    k = sum(1 for d in range(device) if local_cpus[d] == local_cpus[device])      # earlier GPUs on the same socket
    return mine[k % len(mine)]


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
            cpus = _parse_cpulist(open(path).read())
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

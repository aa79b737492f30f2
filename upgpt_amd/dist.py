"""Multi-GPU layer: pure replica data-parallelism over the batch of latents.

Every latent is independent on this path (GroupNorm per (sample, group), attention batched
over (b h), per-sample convs, element-wise DDIM update, per-sample VAE decode — SURVEY.md
§8e), so rank r owns samples [r*B, (r+1)*B), full weights are replicated (0.95 GB fp16 of
288 GB) and there is NO per-step collective.  The single exchange is one all-gather of the
decoded images per batch over RCCL/xGMI (torch.distributed backend "nccl" on ROCm); with
<= 6.3 MB per rank it is latency-, not bandwidth-bound on the 7 x 153 GB/s links.
One process per GPU, rendezvous from the torchrun environment.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def parse_cpulist(text):
    """'0-3,8,10-11' (sysfs cpulist syntax) -> sorted list of CPU numbers."""
    cpus = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return sorted(cpus)


def gpu_numa_cpus(pci_bus_id, sysfs="/sys"):
    """CPUs of the NUMA node the GPU at `pci_bus_id` ("0000:c1:00.0") hangs off, from sysfs, or None when the kernel
    does not say (numa_node = -1, no such device, container without /sys)."""
    try:
        with open(os.path.join(sysfs, "bus", "pci", "devices", pci_bus_id.lower(), "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")) as f:
            cpus = parse_cpulist(f.read())
        return cpus or None
    except (OSError, ValueError):
        return None


def pin_to_gpu_numa(local_rank, sysfs="/sys"):
    """Restricts this process (one rank = one GPU) to the CPUs of its GPU's NUMA node: at N = 8 the eight host
    threads that replay graphs, and RCCL's proxy threads, otherwise migrate across sockets and the launch path of a
    rank crosses the inter-socket link.  Intersected with the CPUs the process may use already (cgroup / taskset);
    a no-op (returns None) when the topology cannot be read or the intersection is empty.  Returns the CPU list."""
    try:
        p = torch.cuda.get_device_properties(local_rank)
        bus = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None
    cpus = gpu_numa_cpus(bus, sysfs)
    if not cpus or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
    if not allowed:
        return None
    os.sched_setaffinity(0, allowed)
    return allowed


def rank_seed(seed, rank):
    """Seed of rank `rank`'s x_T / synthetic inputs (seed + rank; conditioning follows the same seed)."""
    return int(seed) + int(rank)


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* when WORLD_SIZE > 1.
    backend: "nccl" (= RCCL) on GPUs, "gloo" for the CPU tests.  With world > 1 on GPUs the rank is first pinned to
    its GPU's NUMA node (UPGPT_NUMA_PIN=0 switches that off)."""
    rank, local_rank, world = env_world()
    if world > 1 and torch.cuda.is_available() and os.environ.get("UPGPT_NUMA_PIN", "1") == "1":
        pin_to_gpu_numa(local_rank)
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(total, rank, world):
    """Contiguous shard [lo, hi) of `total` samples for `rank` (remainder to the low ranks)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch, rank, world):
    """Slices every tensor / list of a batch dict along dim 0."""
    n = None
    for v in batch.values():
        if torch.is_tensor(v) or isinstance(v, (list, tuple)):
            n = len(v)
            break
    lo, hi = shard_range(n, rank, world)
    return {k: (v[lo:hi] if (torch.is_tensor(v) or isinstance(v, (list, tuple))) and len(v) == n else v)
            for k, v in batch.items()}


def all_gather_images(img, out=None):
    """[b, C, H, W] per rank -> [world*b, C, H, W] on every rank (rank-major order).
    Equal per-rank batch (weak scaling).  No-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return img
    world = dist.get_world_size()
    img = img.contiguous()
    if out is None:
        out = torch.empty((world * img.shape[0],) + tuple(img.shape[1:]), dtype=img.dtype, device=img.device)
    dist.all_gather_into_tensor(out, img)
    return out


def max_over_ranks(value, device):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()

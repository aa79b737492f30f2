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


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* when WORLD_SIZE > 1.
    backend: "nccl" (= RCCL) on GPUs, "gloo" for the CPU tests."""
    rank, local_rank, world = env_world()
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

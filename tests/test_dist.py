"""N > 1 path on CPU: two gloo processes shard a batch, 'compute', and all-gather the images
exactly as bench.py / upgpt_amd.dist do over RCCL on the GPUs."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from upgpt_amd import dist as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    full = {"x_T": torch.arange(6 * 4, dtype=torch.float32).reshape(6, 4), "txt": list("abcdef"), "steps": 50}
    mine = D.shard_batch(full, rank, world)
    lo, hi = D.shard_range(6, rank, world)
    assert torch.equal(mine["x_T"], full["x_T"][lo:hi]) and mine["txt"] == full["txt"][lo:hi] and mine["steps"] == 50
    img = mine["x_T"].reshape(hi - lo, 1, 2, 2) * 2.0  # stand-in for sample+decode of the shard
    allimg = D.all_gather_images(img)
    assert torch.equal(allimg, full["x_T"].reshape(6, 1, 2, 2) * 2.0)
    t = D.max_over_ranks(1.0 + rank, torch.device("cpu"))
    D.barrier()
    q.put((rank, t, tuple(allimg.shape)))
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] == 2.0 for r in res) and all(r[2] == (6, 1, 2, 2) for r in res)


def test_shard_range_covers_everything():
    for total in (1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert D.all_gather_images(torch.ones(2, 3)) is not None  # no process group: identity

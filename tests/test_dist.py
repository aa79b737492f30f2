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


def _bench_worker(rank, world, port, q, backend):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import bench
    D.init_from_env(backend)
    dev = torch.device("cuda", rank) if backend == "nccl" else torch.device("cpu")
    if backend == "nccl":
        torch.cuda.set_device(rank)
    stub = lambda: torch.full((2, 3, 4, 4), float(rank), device=dev)  # stand-in for sample + decode of the rank's batch
    dt, out = bench.timed(bench.gathered(stub), 3, dev)
    dt = D.max_over_ranks(dt, dev)
    ok = out.shape[0] == 2 * world and all(bool((out[2 * r:2 * r + 2] == r).all()) for r in range(world))
    q.put((rank, ok, dt > 0))
    dist.destroy_process_group()


def _run_bench_workers(backend):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True, True), (1, True, True)]


def test_bench_step_gather_and_timing_two_ranks_gloo():
    """bench.py's timed region for N > 1 — step, all-gather of the images, barrier, max over ranks — with a stub
    workload on two gloo ranks."""
    _run_bench_workers("gloo")


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_step_gather_two_ranks_rccl():
    """The same over RCCL (backend "nccl"); needs two GPUs, skipped on the 1-GPU test boxes."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    _run_bench_workers("nccl")


def test_bench_self_launches_its_ranks_and_fails_only_for_lack_of_gpus():
    """`python bench.py --gpus 2` with no torchrun environment re-launches itself under torch.distributed.run (one rank
    per GPU); on a box without GPUs every rank stops at the GPU check — not at a WORLD_SIZE assertion."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    err = r.stderr
    assert "launching 2 ranks" in err and "torch.distributed.run" in err
    assert "needs an MI355X per rank" in err
    assert "AssertionError" not in err and r.returncode != 0


def test_numa_cpulist_and_pinning_lookup(tmp_path):
    """a rank is pinned to the CPUs of its GPU's NUMA node; the sysfs lookup on a fake tree."""
    assert D.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert D.parse_cpulist("") == []
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    (node / "cpulist").write_text("64-127,192-255\n")
    cpus = D.gpu_numa_cpus("0000:C1:00.0", sysfs=str(tmp_path))
    assert cpus[0] == 64 and cpus[-1] == 255 and len(cpus) == 128
    (dev / "numa_node").write_text("-1\n")
    assert D.gpu_numa_cpus("0000:c1:00.0", sysfs=str(tmp_path)) is None  # (the kernel does not know: no pinning)
    assert D.gpu_numa_cpus("0000:ff:00.0", sysfs=str(tmp_path)) is None
    assert D.pin_to_gpu_numa(0, sysfs=str(tmp_path)) is None  # (no GPU here / unknown node: a no-op, never an error)


def test_bench_builds_the_eight_rank_command_and_per_rank_seeds():
    """`python bench.py --gpus 8` re-launches itself as ONE torchrun command with eight ranks on 127.0.0.1 (the
    driver's contract), and rank r draws its x_T / conditioning from seed + r."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    argv = ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    cmd = bench.torchrun_command(8, argv, 29511)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == argv
    assert [D.rank_seed(0, r) for r in range(8)] == list(range(8))
    # different ranks really get different latents, the same rank the same ones
    from upgpt_amd import synth
    a = synth.synth_inputs(2, (8, 8), 4, 87, 768, seed=D.rank_seed(0, 0), text_only=True)["x_T"]
    b = synth.synth_inputs(2, (8, 8), 4, 87, 768, seed=D.rank_seed(0, 1), text_only=True)["x_T"]
    a2 = synth.synth_inputs(2, (8, 8), 4, 87, 768, seed=D.rank_seed(0, 0), text_only=True)["x_T"]
    assert torch.equal(a, a2) and not torch.equal(a, b)

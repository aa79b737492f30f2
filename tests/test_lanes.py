"""Host logic of the execution lanes (upgpt_amd/lanes.py, _lib.lane): step -> lane assignment, per-lane order, the
in-order `after` hook, error propagation, the lane-keyed plan caches, the re-entrant ema_scope, and — world size 2 over
gloo — that an exchange issued from `after` pairs up across ranks whatever the lanes' relative speed."""
import os
import socket
import threading
import time

import pytest
import torch
import torch.multiprocessing as mp

from upgpt_amd import _lib as L
from upgpt_amd.lanes import LanePool, step_lane


def test_lane_context_is_thread_local_and_nests():
    assert L.current_lane() == 0
    seen = {}
    with L.lane(2):
        assert L.current_lane() == 2
        with L.lane(1):
            assert L.current_lane() == 1
        assert L.current_lane() == 2
        t = threading.Thread(target=lambda: seen.setdefault("other", L.current_lane()))
        t.start(); t.join()
    assert L.current_lane() == 0 and seen["other"] == 0
    with pytest.raises(ValueError):
        with L.lane(-1):
            pass


@pytest.mark.parametrize("n,K", [(1, 5), (2, 7), (3, 3), (3, 20), (4, 2)])
def test_pool_runs_step_k_on_lane_k_mod_n_in_lane_order_and_after_in_step_order(n, K):
    pool = LanePool(n, "cpu")
    ran, after_order = [], []
    lock = threading.Lock()

    def fn(k):
        time.sleep(0.002 * ((k * 7) % 3))  # (lanes finish out of step order)
        with lock:
            ran.append((k, L.current_lane()))
        return k * 10

    outs = pool.run(fn, K, after=lambda k, r: after_order.append(k) or r + 1)
    assert outs == [k * 10 + 1 for k in range(K)]
    assert after_order == list(range(K))
    assert sorted(ran) == [(k, step_lane(k, n)) for k in range(K)]
    for i in range(n):  # the steps of one lane ran in order
        mine = [k for k, l in ran if l == i]
        assert mine == sorted(mine)
    assert pool.run(fn, K) == [k * 10 for k in range(K)]


def test_pool_reraises_a_lane_error_on_the_caller():
    pool = LanePool(3, "cpu")

    def fn(k):
        if k == 4:
            raise KeyError("step 4")
        return k

    with pytest.raises(KeyError):
        pool.run(fn, 9)
    assert pool.run(lambda k: k, 3) == [0, 1, 2]  # (the pool is usable afterwards)


def test_ema_scope_is_reentrant_across_threads():
    import upgpt_amd
    m = upgpt_amd.build_model("tiny")
    unet = m.model.diffusion_model
    inside = threading.Barrier(3)
    states = []

    def worker():
        with m.ema_scope():
            inside.wait()
            states.append(unet._weight_override is not None)
            inside.wait()

    th = [threading.Thread(target=worker) for _ in range(2)]
    for t in th:
        t.start()
    inside.wait()          # both workers are inside
    with m.ema_scope():    # nested entry from a third thread
        assert unet._weight_override is not None
    assert unet._weight_override is not None  # the workers still hold it
    inside.wait()
    for t in th:
        t.join()
    assert states == [True, True] and unet._weight_override is None


def _gloo_rank(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from upgpt_amd import dist as D
    pool = LanePool(3, "cpu")

    def fn(k):  # the ranks' lanes run at different speeds
        time.sleep(0.003 * ((k * (5 if rank else 3)) % 4))
        return torch.full((2, 3), float(100 * rank + k))

    outs = pool.run(fn, 8, after=lambda k, img: D.all_gather_images(img))
    q.put((rank, [o[:, 0].tolist() for o in outs]))
    dist.destroy_process_group()


def test_all_gather_from_the_after_hook_pairs_up_across_two_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_gloo_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for r in range(2):  # step k of every rank met step k of the other one
        assert got[r] == [[float(k)] * 2 + [float(100 + k)] * 2 for k in range(8)]


def test_bench_timed_lanes_runs_exactly_n_steps_with_the_exchange_in_step_order():
    """bench.py's timed region in lanes mode (host logic on a CPU pool): exactly n steps, step k on lane k % lanes, the
    exchange hook once per step in step order, the last step's result handed back."""
    import bench
    pool = LanePool(4, "cpu")
    ran, gathered = [], []
    lock = threading.Lock()

    def step(k):
        with lock:
            ran.append((k, L.current_lane()))
        return torch.full((1,), float(k))

    dt, out = bench.timed_lanes(pool, step, 10, "cpu", after=lambda k, img: gathered.append(k) or img + 0.5)
    assert dt > 0 and float(out) == 9.5
    assert sorted(ran) == [(k, k % 4) for k in range(10)] and gathered == list(range(10))


def test_throughput_overlay_is_consulted_only_with_several_batches_in_flight():
    """tuning.TUNE_CACHE_LANES: a well-formed overlay next to the base table, keys of the base table's form, and
    _lib.concurrency() as the switch (set by a LanePool with more than one lane on a GPU; never by a CPU pool)."""
    from upgpt_amd.tuning import TUNE_CACHE, TUNE_CACHE_LANES
    assert TUNE_CACHE_LANES.path.endswith("tuned_gfx950_lanes.json") and len(TUNE_CACHE_LANES.d) >= 20
    assert TUNE_CACHE_LANES.names == TUNE_CACHE.names or TUNE_CACHE_LANES.names is not None
    for k, v in TUNE_CACHE_LANES.d.items():
        assert k.startswith("M") and "_N" in k and len(v) == 4 and int(v[1]) >= 1
    assert L.concurrency() == 1 and L.pools_in_flight() == 1
    with LanePool(3, "cpu") as pool:
        assert L.concurrency() == 1 and pool.n == 3 and L.pools_in_flight() == 1  # (a CPU pool registers nothing)
        with pool.lane(2):
            assert L.concurrency() == 1 and L.current_lane() == 2
    assert L.concurrency() == 1
    # the switch is scoped to the thread and the block: another thread never sees it, leaving the block restores it
    seen = []
    with L.shared_chip(4):
        assert L.concurrency() == 4
        with L.lane(1):  # (a bare lane keeps the enclosing scope's table)
            assert L.concurrency() == 4
        with L.lane(1, concurrency=2):
            assert L.concurrency() == 2
        assert L.concurrency() == 4
        t = threading.Thread(target=lambda: seen.append(L.concurrency()))
        t.start()
        t.join()
    assert seen == [1] and L.concurrency() == 1


def test_a_live_gpu_pool_is_registered_until_closed_and_arms_host_io(monkeypatch):
    """LanePool registers itself process-wide (pools_in_flight: what host_io() looks at besides the thread's own scope)
    and close() / __exit__ / a second close() drop it again; concurrency() of threads outside its lanes stays 1."""
    class FakePool:
        pass
    a, b = FakePool(), FakePool()
    L._register_pool(a, 4)
    L._register_pool(b, 2)
    try:
        assert L.pools_in_flight() == 4 and L.concurrency() == 1
        L._unregister_pool(a)
        assert L.pools_in_flight() == 2
        L._unregister_pool(a)  # (idempotent)
    finally:
        L._unregister_pool(a)
        L._unregister_pool(b)
    assert L.pools_in_flight() == 1


def test_tune_cache_keeps_table_level_decisions_apart_from_shape_entries(tmp_path):
    """"__...__" keys other than "__configs__" (e.g. the row counts whose feed-forward tail runs unfused on a shared chip)
    survive load -> save and never look like shape entries."""
    import json
    from upgpt_amd.tuning import TuneCache, TUNE_CACHE_LANES
    p = tmp_path / "t.json"
    p.write_text(json.dumps({"__configs__": ["a", "b"], "__unfuse_mlp_M__": [8192], "M1_N2_C3+0_k1s1_f0_r000": [1, 1, 2.0, 3.0]}))
    c = TuneCache(str(p))
    assert c.meta == {"__unfuse_mlp_M__": [8192]} and list(c.d) == ["M1_N2_C3+0_k1s1_f0_r000"] and c.names == ["a", "b"]
    c.save(str(tmp_path / "u.json"))
    back = json.loads((tmp_path / "u.json").read_text())
    assert back["__unfuse_mlp_M__"] == [8192] and back["__configs__"] == ["a", "b"] and "M1_N2_C3+0_k1s1_f0_r000" in back
    assert 8192 in TUNE_CACHE_LANES.meta.get("__unfuse_mlp_M__", [])


def test_host_io_is_free_with_one_batch_in_flight_and_exclusive_with_several():
    order = []
    with L.host_io():  # (concurrency 1: nothing to take)
        with L.host_io():
            order.append("nested")
    class FakePool:
        pass
    pool = FakePool()
    L._register_pool(pool, 4)  # (a live pool arms the lock for every thread of the process)
    try:
        inside = threading.Event()
        release = threading.Event()

        def holder():
            with L.host_io():
                inside.set()
                release.wait(5)
                order.append("holder out")

        def waiter():
            inside.wait(5)
            with L.host_io():
                order.append("waiter in")

        ts = [threading.Thread(target=holder), threading.Thread(target=waiter)]
        for t in ts:
            t.start()
        inside.wait(5)
        time.sleep(0.05)
        assert "waiter in" not in order  # the second thread is held at the lock
        release.set()
        for t in ts:
            t.join(5)
        assert order == ["nested", "holder out", "waiter in"]
    finally:
        L._unregister_pool(pool)

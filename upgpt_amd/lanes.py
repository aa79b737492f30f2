"""Several independent batches in flight on one MI355X.

At bs = 8 the UNet forward is a chain of ~250 dependent launches of ~11 us, each with one tile per CU: the matrix
pipes are busy ~3 % of the time and every launch pays its boundary, its first memory round trips and the drain of its
stores.  Nothing inside ONE chain can use that slack — but a second, independent batch can: its
kernels are dispatched from another HIP stream into the boundaries, prologues and tails of the first one's.  A LANE
is what makes two batches independent on the device: its own upk_ctx (split-K workspace), its own plans (activation
buffers, captured step graphs), its own stream and its own host thread.  The packed weights are shared, every
sample() call is still one bs = B batch with the reference's call surface, and a lane's results are bit-identical to
the serial path's (tests/test_lanes_gpu.py).  This is the serving shape of the path — independent requests of one
batch each — not a larger batch: the batch of a UNetModel.forward does not change.

Plans built INSIDE a lane of a pool with more than one lane (LanePool.lane(i): the pool's own threads, or a script's main
thread that enters it) take the launch choices tuned for a shared chip (tuning.TUNE_CACHE_LANES: chosen by chip time with
four chains in flight — few large tiles instead of one small tile per CU); the switch is scoped to the thread inside the
lane and part of every plan key (_lib.concurrency), so callers outside the pool keep the single-forward table.  While such a
pool is alive (until close() / __exit__), graph captures / plan construction / uploads from pageable host memory of ALL
threads are serialised against each other (_lib.host_io; a steady-state sample() with its inputs on the device uploads
nothing).  DESIGN.md 13.
"""
import contextlib
import os
import threading
import time

import torch

from ._check import require
from ._lib import lane


def step_lane(k, n_lanes):
    """Lane that runs step k of a job: round robin."""
    return k % n_lanes


# ---- CU-partitioned lane streams: a lane's kernels only on its own CUs
# What the CU mask of an HSA queue does on this driver, measured (scripts/r6_lanes_lab.py maskdiag,
# profiles/r06_cu_mask_semantics.txt): mask bit b addresses CU slot b // 8 of XCD b % 8, and an XCD whose bits are ALL zero
# is unrestricted.  A stream can therefore not be confined to a subset of the XCDs (the rationale of the experiment:
# a lane's activations inside two L2s); what can be built is a partition by CU SLOT: lane l gets slots [l * 32 / n,
# (l + 1) * 32 / n) of every XCD — disjoint CUs, all eight L2s shared as before.
N_XCD, N_CU = 8, 256


def cu_slot_mask_words(slots):
    """CU mask (N_CU bits as 32-bit words) allowing CU slots `slots` (0..31) of every XCD: bit b <-> slot b // 8 of XCD b % 8."""
    words = [0] * (N_CU // 32)
    for b in range(N_CU):
        if b // N_XCD in slots:
            words[b // 32] |= 1 << (b % 32)
    return words


def probe_placement(ctx, stream, nblocks=2048, spin=40000):
    """-> {xcd: number of distinct CUs} the workgroups of a launch on `stream` ran on (upk_probe_placement)."""
    import ctypes as C
    out = torch.zeros(nblocks * 2, dtype=torch.int32, device=ctx.device)
    ptr = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
    ctx._chk(ctx.lib.upk_probe_placement(ctx.h, out.data_ptr(), nblocks, spin, C.c_void_p(ptr)))
    torch.cuda.synchronize(ctx.device)
    v = out.cpu().view(nblocks, 2)
    seen = {}
    for x, cu in zip((v[:, 0] & 0xF).tolist(), ((v[:, 1] >> 8) & 0xFF).tolist()):  # HW_ID[15:8] = {se_id, sh_id, cu_id}
        seen.setdefault(x, set()).add(cu)
    return {x: len(c) for x, c in sorted(seen.items())}


def cu_partition_streams(ctx, n):
    """n HIP streams on disjoint CU sets: stream l may use CU slots [l * 32 // n, (l + 1) * 32 // n) of every XCD
    (upk_stream_create_cumask), wrapped as torch.cuda.ExternalStream.  Returns (streams, [placement of the probe kernel per
    stream]); raises when the probe finds a stream on more CUs than its mask allows (another mask convention)."""
    import ctypes as C
    require(1 <= n <= 32 and 32 % n == 0, "CU partitions: n must divide the 32 CU slots of an XCD", ValueError)
    streams, seen = [], []
    for l in range(n):
        words = cu_slot_mask_words(set(range(l * 32 // n, (l + 1) * 32 // n)))
        arr = (C.c_uint32 * len(words))(*words)
        h = C.c_void_p()
        ctx._chk(ctx.lib.upk_stream_create_cumask(ctx.h, arr, len(words), C.byref(h)))
        s = torch.cuda.ExternalStream(h.value, device=ctx.device)
        streams.append(s)
        seen.append(probe_placement(ctx, s))
        require(sum(seen[-1].values()) <= N_CU // n, lambda: "CU mask not honoured as expected: %r" % (seen[-1],), RuntimeError)
    return streams, seen


class LanePool:
    """`n` execution lanes on one device.  run(fn, K) executes fn(step) for step = 0 .. K-1, step k on lane
    k % n, the steps of a lane in order on that lane's stream and host thread; `after(step, result)` — if given —
    is called on the CALLING thread, in step order, on a stream that waits for the result (the place for an exchange
    whose order must be the same on every rank: the all-gather of the images).  device "cpu" keeps the threading and the
    ordering and drops the streams (host-logic tests; the HIP path itself has no CPU form)."""

    def __init__(self, n, device=None):
        require(int(n) >= 1, "LanePool needs at least one lane", ValueError)
        self.n = int(n)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.gpu = self.device.type == "cuda"
        self._closed = False
        if self.gpu and self.n > 1:
            from ._lib import _register_pool, get_context
            _register_pool(self, self.n)  # (host_io() serialises captures against uploads while the pool lives)
            for i in range(self.n):  # every lane's upk_ctx + workspace exists before the first thread starts
                get_context(self.device, lane=i)
        self.queue_probe = "n/a"
        # a one-lane pool is exactly the serial path (caller's stream); with more lanes every lane has a stream of its own
        # (UPGPT_LANE0_MAIN=1: lane 0 on the caller's stream instead)
        self.lane0_main = os.environ.get("UPGPT_LANE0_MAIN", "0") == "1"
        k = self.n - 1 if self.lane0_main else self.n
        self.streams = [None] * self.n if not self.gpu or self.n == 1 else (
            ([None] if self.lane0_main else []) + self._distinct_queue_streams(k, with_main=self.lane0_main))
        self._xstream = torch.cuda.Stream(device=self.device) if self.gpu and self.n > 1 else None

    # ---- stream -> hardware queue placement
    def _overlap(self, sa, sb, cycles):
        """Do kernels on streams sa and sb run at the same time?  One spinning single-thread kernel on each: two streams
        that the HIP runtime mapped to ONE hardware queue run them back to back."""
        def run(streams):
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for s in streams:
                with torch.cuda.stream(s):
                    torch.cuda._sleep(cycles)
            torch.cuda.synchronize(self.device)
            return time.perf_counter() - t0
        one = min(run([sa]) for _ in range(3))
        two = min(run([sa, sb]) for _ in range(3))
        return two < 1.5 * one, one

    def _distinct_queue_streams(self, k, candidates=12, with_main=True):
        """k streams that share a hardware queue neither with the caller's stream nor with each other.  The runtime
        multiplexes its streams onto a few hardware queues (4 by default) in creation order; two lanes that land on one
        queue take turns instead of overlapping (measured: 5.70 ms for two forwards against 4.25 ms on two queues,
        scripts/stream_queues.py), so the pool measures instead of trusting the order.  Falls back to fresh streams when
        the probe is unavailable or finds too few queues."""
        with torch.cuda.device(self.device):
            main = torch.cuda.current_stream(self.device)
            cands = [torch.cuda.Stream(device=self.device) for _ in range(candidates)]
            chosen = []
            try:
                cycles = 200000
                ok, one = self._overlap(main, main, cycles)  # (calibration: the same stream never overlaps itself)
                if ok or one < 2e-5:
                    raise RuntimeError("spin probe does not resolve on this device")
                for c in cands:
                    if len(chosen) == k:
                        break
                    if all(self._overlap(o, c, cycles)[0] for o in ([main] if with_main else []) + chosen):
                        chosen.append(c)
            except Exception:
                chosen = []
            self.queue_probe = "measured" if len(chosen) == k else "unverified"
            if len(chosen) < k:
                chosen = (chosen + [c for c in cands if c not in chosen])[:k]
            return chosen

    def _lane_steps(self, i, stream, fn, K, slots, cond, errors):
        try:
            with (torch.cuda.device(self.device) if self.gpu else contextlib.nullcontext()), self.lane(i, stream):
                prev = None
                if self.gpu:
                    prev = torch.cuda.Event(enable_timing=True)
                    prev.record(torch.cuda.current_stream(self.device))
                for k in range(i, K, self.n):
                    out = fn(k)
                    ev = None
                    if self.gpu:
                        ev = torch.cuda.Event(enable_timing=True)
                        ev.record(torch.cuda.current_stream(self.device))
                        self._spans.append((i, k, prev, ev))  # (device time of step k on its lane: lane_step_ms())
                        prev = ev
                    with cond:
                        slots[k] = (out, ev)
                        cond.notify_all()
        except BaseException as e:  # (re-raised on the calling thread)
            with cond:
                errors.append(e)
                cond.notify_all()

    def lane(self, i, stream=None):
        """Context manager: the calling thread works in lane `i` of THIS pool — the lane's stream (unless `stream` is given)
        and plans tuned for self.n batches sharing the chip.  The pool's own threads run in it; a script that builds or
        replays a lane's plans from its main thread enters it too.  Nothing outside the block changes: a thread that never
        enters a lane keeps the single-forward launch choices while the pool exists."""
        require(0 <= int(i) < self.n, "lane index out of range", ValueError)
        s = stream if stream is not None else self.streams[int(i)]
        return lane(i, s, concurrency=self.n if (self.gpu and self.n > 1) else 1)

    def close(self):
        """The pool stops counting as in flight (the lanes' contexts, plans and graphs stay cached for the next pool).
        Idempotent; also run by __exit__ and __del__."""
        if not self._closed:
            self._closed = True
            from ._lib import _unregister_pool
            _unregister_pool(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def lane_step_ms(self):
        """Device time of every step of the last run() on its lane, per lane: [[ms, ...] for each lane] (synchronises)."""
        out = [[] for _ in range(self.n)]
        if self.gpu:
            torch.cuda.synchronize(self.device)
            for i, k, e0, e1 in sorted(getattr(self, "_spans", []), key=lambda t: t[1]):
                out[i].append(e0.elapsed_time(e1))
        return out

    def run(self, fn, K, after=None):
        """-> list of the K results (of `after` when given).  Device work may still be in flight when run() returns; the
        calling thread's current stream has been made to wait for all of it."""
        if self.n == 1:
            outs = []
            for k in range(K):
                out = fn(k)
                outs.append(after(k, out) if after is not None else out)
            return outs
        main = torch.cuda.current_stream(self.device) if self.gpu else None
        slots, errors, cond = [None] * K, [], threading.Condition()
        self._spans = []
        streams = [main if s is None else s for s in self.streams]
        if self.gpu:
            for s in streams:
                if s is not main:
                    s.wait_stream(main)  # (inputs prepared on the caller's stream are visible to every lane)
        threads = [threading.Thread(target=self._lane_steps, args=(i, streams[i], fn, K, slots, cond, errors), daemon=True)
                   for i in range(self.n)]
        for t in threads:
            t.start()
        outs = [None] * K
        # `after` runs on a stream of its own: a wait put into a lane's stream would stall that lane behind another one
        xs = self._xstream if after is not None else None
        try:
            for k in range(K):
                with cond:
                    while slots[k] is None and not errors:
                        cond.wait()
                    if errors:
                        break
                    out, ev = slots[k]
                if after is None:
                    if self.gpu and torch.is_tensor(out):
                        out.record_stream(main)  # (allocated on its lane's stream, read by the caller on this one)
                    outs[k] = out
                    continue
                if not self.gpu:
                    outs[k] = after(k, out)
                    continue
                xs.wait_event(ev)
                if torch.is_tensor(out):
                    out.record_stream(xs)
                with torch.cuda.stream(xs):
                    res = outs[k] = after(k, out)
                if torch.is_tensor(res):
                    res.record_stream(main)
        finally:
            for t in threads:
                t.join()
            if self.gpu:
                for s in streams + ([xs] if xs is not None else []):
                    if s is not main:
                        main.wait_stream(s)
        if errors:
            raise errors[0]
        return outs

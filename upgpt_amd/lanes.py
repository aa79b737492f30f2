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
"""
import contextlib
import threading

import torch

from ._check import require
from ._lib import lane


def step_lane(k, n_lanes):
    """Lane that runs step k of a job: round robin."""
    return k % n_lanes


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
        # lane 0 runs on the caller's stream, so that a one-lane pool is exactly the serial path
        self.streams = [None] + [torch.cuda.Stream(device=self.device) if self.gpu else None for _ in range(self.n - 1)]
        self._xstream = torch.cuda.Stream(device=self.device) if self.gpu and self.n > 1 else None

    def _lane_steps(self, i, stream, fn, K, slots, cond, errors):
        try:
            with (torch.cuda.device(self.device) if self.gpu else contextlib.nullcontext()), lane(i, stream):
                for k in range(i, K, self.n):
                    out = fn(k)
                    ev = None
                    if self.gpu:
                        ev = torch.cuda.Event()
                        ev.record(torch.cuda.current_stream(self.device))
                    with cond:
                        slots[k] = (out, ev)
                        cond.notify_all()
        except BaseException as e:  # (re-raised on the calling thread)
            with cond:
                errors.append(e)
                cond.notify_all()

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
        streams = [main] + self.streams[1:]
        if self.gpu:
            for s in streams[1:]:
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
                for s in streams[1:] + ([xs] if xs is not None else []):
                    main.wait_stream(s)
        if errors:
            raise errors[0]
        return outs

"""Experiment: L independent bs = 8 batches in flight on one GPU (L execution lanes: own upk_ctx / split-K workspace /
activation buffers / graphs / stream, shared packed weights; one host thread per lane) against the serial loop.
Each lane runs whole bench steps: 50-step DDIM sample() + VAE decode.  Usage: multibatch.py [steps per lane] [lanes ...]"""
import contextlib, io, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import upgpt_amd
from upgpt_amd import synth
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
LANES = [int(v) for v in sys.argv[2:]] or [1, 2, 3, 4]
hw = tuple(int(v) for v in os.environ.get("LATENT", "32x32").split("x"))
decode = os.environ.get("DECODE", "1") == "1"


def lane_loop(i, wl, stream, n, outs):
    with upgpt_amd.lane(i, stream):
        out = None
        for _ in range(n):
            z, _ = wl.sampler.sample(wl.S, wl.B, (4,) + tuple(wl.hw), wl.cond, eta=0.0, x_T=wl.x_T, verbose=False, log_every_t=10 ** 6)
            out = model.decode_first_stage(z) if decode else z
        outs[i] = out


def run(L, wls, streams, n):
    outs = [None] * L
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with model.ema_scope(), contextlib.redirect_stdout(io.StringIO()):
        if L == 1:
            lane_loop(0, wls[0], streams[0], n, outs)
        else:
            th = [threading.Thread(target=lane_loop, args=(i, wls[i], streams[i], n, outs)) for i in range(L)]
            for t in th: t.start()
            for t in th: t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, outs


from upgpt_amd.lanes import LanePool
wls, streams = [], []
for L in LANES:
    while len(wls) < L:
        wls.append(bench.Workload(model, 8, hw, 50, seed=len(wls)))
    pool = LanePool(L)  # (streams probed to sit on distinct hardware queues: unprobed ones gave 60-80 img/s at random)
    streams = [s if s is not None else torch.cuda.current_stream() for s in pool.streams]
    for i in range(L):  # warm every lane serially (plans, tuning lookups, graphs)
        run(1, [wls[i]], [streams[i]], 1) if i == 0 else None
    with model.ema_scope(), contextlib.redirect_stdout(io.StringIO()):
        for i in range(L):
            lane_loop(i, wls[i], streams[i], 1, [None] * L)
    torch.cuda.synchronize()
    best = None
    for rep in range(3):
        dt, outs = run(L, wls, streams, K)
        best = dt if best is None else min(best, dt)
    cs = [float(o.double().sum()) for o in outs[:L]]
    print("lanes %d: %d bench steps (%d per lane) in %.1f ms -> %.2f ms per step, %.2f img/s   checksums %s" % (
        L, L * K, K, best * 1e3, best / (L * K) * 1e3, 8 * L * K / best, " ".join("%.4f" % c for c in cs)), flush=True)

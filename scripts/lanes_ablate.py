"""Where the time goes with L forwards in flight: the captured UNet forward of every lane replayed concurrently (probed
streams), in full and without one class / one group of launches at a time; the difference per forward is what that group
costs in THROUGHPUT terms (chip time), next to its serial cost.  Usage: lanes_ablate.py [lanes]"""
import contextlib, ctypes as C, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd
from upgpt_amd import synth
from upgpt_amd.lanes import LanePool
LANES = int(sys.argv[1]) if len(sys.argv) > 1 else 4
REPS = 8
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
unet = model.model.diffusion_model
pool = LanePool(LANES)
print("lanes", LANES, "queues", pool.queue_probe, flush=True)
plans = []
for i in range(LANES):
    with pool.lane(i):
        p = unet.plan(8, 32, 32, 87, 50, "sampler"); p.prep.run(); plans.append(p)
torch.cuda.synchronize()
streams = [s if s is not None else torch.cuda.current_stream() for s in pool.streams]


def replay(n_lanes, skip=(), skip_idx=()):
    gs = []
    for p, s in list(zip(plans, streams))[:n_lanes]:
        ctx = p.ctx
        with torch.cuda.stream(s):
            ctx._chk(ctx.lib.upk_graph_begin(ctx.h, s.cuda_stream))
            p.body.run(s.cuda_stream, skip=skip, skip_idx=skip_idx)
            g = C.c_void_p(); ctx._chk(ctx.lib.upk_graph_end(ctx.h, s.cuda_stream, C.byref(g)))
        gs.append(g)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(REPS):
            for p, g, s in zip(plans, gs, streams):
                p.ctx._chk(p.lib.upk_graph_launch(p.hctx, g, s.cuda_stream))
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (REPS * n_lanes) * 1e3)
    for p, g in zip(plans, gs):
        p.ctx.graph_destroy(g)
    return best


body = plans[0].body
full1, fullL = replay(1), replay(LANES)
print("forward: %.3f ms alone, %.3f ms per forward with %d in flight" % (full1, fullL, LANES), flush=True)
groups = {}
for i, (c, lab) in enumerate(zip(body.cls, body.labels)):
    groups.setdefault(c, []).append(i)
    if c.startswith("igemm"):
        key = "  " + (lab.split("_C")[0] + ("_k3" if "_k3" in lab else "_k1") if lab.startswith("M") else lab.split(" M")[0])
        groups.setdefault(key, []).append(i)
    elif c == "attention":
        groups.setdefault("  " + lab, []).append(i)
print("%-44s %5s %10s %10s" % ("group", "ops", "serial us", "lanes us"))
for k, idx in groups.items():
    a = (full1 - replay(1, skip_idx=frozenset(idx))) * 1e3
    b = (fullL - replay(LANES, skip_idx=frozenset(idx))) * 1e3
    print("%-44s %5d %10.1f %10.1f" % (k, len(idx), a, b), flush=True)

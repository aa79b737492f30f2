"""Experiment: one HIP graph whose two parallel branches each run a B=4 UNet body (fork/join by
events during capture) vs a single B=8 body."""
import contextlib, io, os, sys, time
import ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd
from upgpt_amd import synth
from upgpt_amd.engine import UNetPlan
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
unet = model.model.diffusion_model
ctx, tag, pk = unet.packed()
H = W = 32
def capture(plans):
    main = torch.cuda.Stream()
    sides = [torch.cuda.Stream() for _ in plans[1:]]
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        ctx._chk(ctx.lib.upk_graph_begin(ctx.h, main.cuda_stream))
        ev0 = torch.cuda.Event(); ev0.record(main)
        evs = []
        for s, p in zip(sides, plans[1:]):
            s.wait_event(ev0)
            p.body.run(s.cuda_stream)
            e = torch.cuda.Event(); e.record(s); evs.append(e)
        plans[0].body.run(main.cuda_stream)
        for e in evs: main.wait_event(e)
        g = C.c_void_p(); ctx._chk(ctx.lib.upk_graph_end(ctx.h, main.cuda_stream, C.byref(g)))
    return main, g
def bench(main, g, n=20):
    ctx._chk(ctx.lib.upk_graph_launch(ctx.h, g, main.cuda_stream)); main.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): ctx._chk(ctx.lib.upk_graph_launch(ctx.h, g, main.cuda_stream))
    main.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for nb in (1, 2, 4):
    B = 8 // nb
    plans = []
    for i in range(nb):
        p = UNetPlan(ctx, pk, B, H, W, 87, 50, "sampler"); p.apply_tuning(); p.prep.run(); plans.append(p)
    main, g = capture(plans)
    print("%d parallel branch(es) of B=%d in ONE graph: %.3f ms per forward of 8 samples" % (nb, B, bench(main, g)), flush=True)

"""Experiment: B=8 on one stream vs k sub-batches on k parallel streams (graph replays interleaved)."""
import contextlib, io, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd
from upgpt_amd import synth
from upgpt_amd.engine import SamplerState
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
unet = model.model.diffusion_model
H = W = 32
S = 50
for nsub in (1, 2, 4, 8):
    B = 8 // nsub
    pl = unet.plan(B, H, W, 87, S, "sampler")
    pl.prep.run()
    streams = [torch.cuda.Stream() for _ in range(nsub)]
    # one SamplerState per sub-batch needs its own buffers: build independent plans by bumping rows key trick
    plans = [pl]
    from upgpt_amd.engine import UNetPlan
    ctx, tag, pk = unet.packed()
    for i in range(1, nsub):
        p2 = UNetPlan(ctx, pk, B, H, W, 87, S, "sampler"); p2.apply_tuning(); p2.prep.run(); plans.append(p2)
    states = [SamplerState(p, 4) for p in plans]
    torch.cuda.synchronize()
    graphs = []
    for st, s in zip(states, streams):
        with torch.cuda.stream(s):
            graphs.append(st.graph(False))
    torch.cuda.synchronize()
    def run():
        for st in states: st.plan.step.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(S):
            for st, s, g in zip(states, streams, graphs):
                ctx._chk(ctx.lib.upk_graph_launch(ctx.h, g, s.cuda_stream))
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    run()
    t = min(run() for _ in range(3))
    print("sub-batches %d x B=%d on %d streams: %.1f ms per 50 steps  (%.2f ms/step, %.1f img/s UNet-only)" % (nsub, B, nsub, t * 1e3, t / S * 1e3, 8 / t), flush=True)
    del plans, states, graphs

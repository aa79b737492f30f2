"""Diagnostic: which HIP streams share a hardware queue?  Two captured UNet forwards (two lanes) replayed concurrently
on every pair of candidate streams: a pair on one hardware queue serialises (2 x 2.85 ms), a pair on two queues overlaps."""
import contextlib, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd
from upgpt_amd import synth
from upgpt_amd.engine import SamplerState
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
unet = model.model.diffusion_model
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
plans, states = [], []
for i in range(2):
    with upgpt_amd.lane(i):
        p = unet.plan(8, 32, 32, 87, 50, "sampler"); p.prep.run()
        plans.append(p); states.append(SamplerState(p, 4))
torch.cuda.synchronize()
graphs = []
for i, st in enumerate(states):
    with upgpt_amd.lane(i):
        graphs.append(st.graph(False))
torch.cuda.synchronize()
streams = [("null", torch.cuda.default_stream())] + [("s%d" % k, torch.cuda.Stream()) for k in range(N)]
streams += [("hi%d" % k, torch.cuda.Stream(priority=-1)) for k in range(2)]
print("streams:", " ".join("%s=%#x" % (n, s.cuda_stream) for n, s in streams), flush=True)


def pair(sa, sb, reps=6):
    best = 1e9
    for _ in range(3):
        for p in plans: p.step.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for p, g, s in zip(plans, graphs, (sa, sb)):
                p.ctx._chk(p.lib.upk_graph_launch(p.hctx, g, s.cuda_stream))
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best


print("      " + " ".join("%6s" % n for n, _ in streams))
for na, sa in streams:
    print("%6s" % na + " " + " ".join("%6.2f" % pair(sa, sb) if na != nb else "   -  " for nb, sb in streams), flush=True)

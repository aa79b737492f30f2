"""Replays the captured UNet forward (B = 8, 32 x 32 latent, shared-chip tuning table) of L lanes concurrently, N times per
lane — the timed configuration of bench.py without the host side: target for rocprofv3 passes (--kernel-trace, --pmc).
  python scripts/lanes_replay.py [lanes=4] [forwards per lane=6] [table: lanes the plans are tuned for, default = lanes]"""
import contextlib, ctypes as C, io, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd
from upgpt_amd import synth
from upgpt_amd.engine import SamplerState
from upgpt_amd.lanes import LanePool
LANES = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
TABLE = int(sys.argv[3]) if len(sys.argv) > 3 else LANES
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
unet = model.model.diffusion_model
pool = LanePool(LANES)
plans, states, graphs = [], [], []
for i in range(LANES):
    inp = synth.synth_inputs(8, (32, 32), 4, 87, 768, seed=1000 * i, text_only=True)
    with upgpt_amd._lib.lane(i, pool.streams[i], concurrency=TABLE):
        pl = unet.plan(8, 32, 32, 87, 50, "sampler")
        pl.load_x_nchw(inp["x_T"].cuda(), 0, 0); pl.load_x_nchw(inp["c_concat"].cuda(), 4, pl.cin_pad)
        pl.load_context(inp["c_crossattn"].cuda()); pl.t_rows.copy_(torch.arange(981, 0, -20, dtype=torch.float32)[:50])
        pl._t_rows_key = None
        pl.prep.run()
        st = SamplerState(pl, 4); st.x.copy_(inp["x_T"].cuda()); st.coefs.fill_(0.5)
        graphs.append(st.graph(False))
    plans.append(pl); states.append(st)
torch.cuda.synchronize()
streams = [s if s is not None else torch.cuda.current_stream() for s in pool.streams]
for r in range(N):
    for p, g, s in zip(plans, graphs, streams):
        if r % 40 == 0:  # the device-side step index addresses a 50-row table: back to row 0 before it runs out
            with torch.cuda.stream(s):
                p.step.zero_()
        p.ctx._chk(p.lib.upk_graph_launch(p.hctx, g, s.cuda_stream))
torch.cuda.synchronize()
print("replayed %d forwards on each of %d lanes (%s queues); conv/GEMM launches per forward: %d" % (
    N, LANES, pool.queue_probe, sum(1 for c in plans[0].body.cls if c.startswith("igemm"))))

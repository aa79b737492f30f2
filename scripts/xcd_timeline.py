"""Dev tool: in-kernel clock stamps of the per-XCD engine's phases for ONE transformer block inside an eager forward.
  python scripts/xcd_timeline.py [op-ordinal of the xcd op, default 3] [H W]"""
import contextlib, io, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("UPGPT_XCD_MAXN", "1024")
import upgpt_amd
from upgpt_amd import synth
which = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
W = int(sys.argv[3]) if len(sys.argv) > 3 else 32
B = int(os.environ.get("TL_B", "8"))
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
unet = model.model.diffusion_model
inp = synth.synth_inputs(B, (H, W), 4, 87, 768, seed=0, text_only=True)
pl = unet.plan(B, H, W, 87, 50, "sampler")
pl.load_x_nchw(inp["x_T"].cuda(), 0, 0); pl.load_x_nchw(inp["c_concat"].cuda(), 4, pl.cin_pad)
pl.load_context(inp["c_crossattn"].cuda()); pl.t_rows.copy_(torch.arange(981, 0, -20, dtype=torch.float32)[:50])
pl.prep.run()
ctx = pl.ctx
NPH = 10
buf = torch.zeros(256 * NPH * 8, dtype=torch.int64, device="cuda")
xs = [i for i, l in enumerate(pl.body.labels) if l.startswith("xcd ")]
tgt = xs[which]
names = ["gn", "proj_in", "qkv", "attn1", "out1", "q2", "attn2", "out2", "geglu", "ffout"]
for rep in range(3):
    s = ctx._s()
    for i, op in enumerate(pl.body.ops):
        if i == tgt:
            ctx.lib.upk_xcd_dev_timeline(buf.data_ptr())
        op(s)
        if i == tgt:
            ctx.lib.upk_xcd_dev_timeline(None)
    torch.cuda.synchronize()
t = buf.view(256, NPH, 8).cpu().double()
act0 = t[:, 0, 5] > 0
t0 = t[act0, 0, 0].min()
print(pl.body.labels[tgt], "(op %d); shader-clock ticks; per phase the SLOWEST workgroup's deltas" % tgt)
print("%-8s %9s | %7s %7s %7s %7s %7s %7s | %8s" % ("phase", "start", "desc", "A-loads", "A->LDS", "K loop", "epilog", "barrier", "total"))
for ph in range(NPH):
    a = t[:, ph, :]
    act = a[:, 5] > 0
    if not bool(act.any()):
        continue
    a = a[act]
    g = a[:, 1] > 0  # GEMM phases carry the inner stamps
    def mx(x):
        return float(x.max()) if x.numel() else 0.0
    if bool(g.any()):
        ag = a[g]
        cols = (mx(ag[:, 1] - ag[:, 0]), mx(ag[:, 2] - ag[:, 1]), mx(ag[:, 3] - ag[:, 2]), mx(ag[:, 4] - ag[:, 3]), mx(ag[:, 5] - ag[:, 4]))
    else:
        cols = (0, 0, 0, mx(a[:, 5] - a[:, 0]), 0)
    print("%-8s %9.0f | %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f | %8.0f" % ((names[ph], float(a[:, 0].min() - t0)) + cols + (
        float((a[:, 6] - a[:, 5]).min()), float(a[:, 6].max() - a[:, 0].min()))))
print("whole launch: %.0f ticks" % float(t[act0, NPH - 1, 6].max() - t0))

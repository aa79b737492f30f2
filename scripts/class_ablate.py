"""True per-class time of one UNet forward under HIP-graph replay: capture the body with one
class of launches removed and subtract (dev tool; results of ablated runs are garbage)."""
import contextlib, io, os, sys, time
import ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd
from upgpt_amd import synth
H = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W = int(sys.argv[2]) if len(sys.argv) > 2 else 32
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
unet = model.model.diffusion_model
pl = unet.plan(8, H, W, 87, 50, "sampler")
pl.prep.run(); torch.cuda.synchronize()
ctx = pl.ctx
def timed(skip):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        sp = s.cuda_stream
        ctx._chk(ctx.lib.upk_graph_begin(ctx.h, sp))
        pl.body.run(sp, skip=skip)
        g = C.c_void_p(); ctx._chk(ctx.lib.upk_graph_end(ctx.h, sp, C.byref(g)))
        ctx._chk(ctx.lib.upk_graph_launch(ctx.h, g, sp)); s.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): ctx._chk(ctx.lib.upk_graph_launch(ctx.h, g, sp))
        s.synchronize()
        return (time.perf_counter() - t0) / 10 * 1e3
from collections import Counter
cnt = Counter(pl.body.cls)
full = timed(())
print("full forward %.3f ms  launches by class: %s" % (full, dict(cnt)))
for c in [int(v) for v in os.environ.get("FORCE_CFGS", "").split(",") if v]:
    # one igemm binary for every launch (split-K by the cost model): code-locality experiment
    ctx.conv_override(c, 0)
    print("  every igemm forced to cfg %s: %.3f ms" % (ctx.lib.upk_conv_config_name(c).decode(), timed(())))
    ctx.conv_override(-1, 0)
if os.environ.get("FORCE_ONLY"): sys.exit(0)
for cls in sorted(cnt):
    t = timed((cls,))
    print("  without %-12s %.3f ms  -> class costs %.3f ms (%d launches, %.1f us each)" % (cls, t, full - t, cnt[cls], (full - t) / cnt[cls] * 1e3))

"""Which producer launches of the UNet plan leave LayerNorm row sums (upk_conv_ln_rows), by shape and tuned tile
configuration (dev tool)."""
import contextlib, io, os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd
from upgpt_amd import synth
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
unet = model.model.diffusion_model
pl = unet.plan(8, 32, 32, 87, 50, "sampler")
ctx = pl.ctx
seen = {}
for d, key in pl.convs:
    if d.ln_rows_out:
        s = C.c_int(-1)
        ctx._chk(ctx.lib.upk_conv_ln_rows(ctx.h, C.byref(d), C.byref(s)))
        name = ctx.lib.upk_conv_config_name(d.tune_cfg - 1).decode() if d.tune_cfg > 0 else "?"
        seen.setdefault((key, name, d.tune_splitk, s.value), 0)
        seen[(key, name, d.tune_splitk, s.value)] += 1
for k, n in sorted(seen.items()):
    print(n, k)

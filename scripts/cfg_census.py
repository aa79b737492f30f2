"""Which tile configuration every conv/GEMM launch of the bench forward runs on (dev tool)."""
import contextlib, io, os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd
from upgpt_amd import synth
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
unet = model.model.diffusion_model
pl = unet.plan(8, 32, 32, 87, 50, "sampler")
lib = pl.ctx.lib
body = set(id(k) for k in pl.body.keep)
cnt = collections.Counter()
for d, key in pl.convs:
    if id(d) in body:
        nm = lib.upk_conv_config_name(d.tune_cfg - 1).decode() if d.tune_cfg > 0 else "auto"
        cnt[(nm, d.tune_splitk)] += 1
        print("%-52s %-14s sk %d" % (key, nm, d.tune_splitk))
print()
for k, v in cnt.most_common():
    print(v, k)

#!/bin/bash
# A/B on one box: UNet forward (graph replay, B=AB_B (8), latent AB_HW=32x32) under VAR=value for each value, interleaved twice.
#   bash scripts/ab_env.sh UPGPT_GN_REDUCE_APPLY 0 1
R=$GRAFT_REPO_ROOT; cd $R
VAR=$1; shift
for v in "$@" "$@"; do
  env $VAR=$v AB_VAR=$VAR timeout 300 python - <<'PY' 2>/dev/null | tail -1
import contextlib, io, os, sys, time, torch
sys.path.insert(0, os.getcwd())
import upgpt_amd
from upgpt_amd import synth
from upgpt_amd.engine import SamplerState
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
unet = model.model.diffusion_model
HW = tuple(int(v) for v in os.environ.get("AB_HW", "32x32").split("x"))
B = int(os.environ.get("AB_B", "8"))
inp = synth.synth_inputs(B, HW, 4, 87, 768, seed=0, text_only=True)
pl = unet.plan(B, HW[0], HW[1], 87, 50, "sampler")
pl.load_x_nchw(inp["x_T"].cuda(), 0, 0); pl.load_x_nchw(inp["c_concat"].cuda(), 4, pl.cin_pad)
pl.load_context(inp["c_crossattn"].cuda()); pl.t_rows.copy_(torch.arange(981, 0, -20, dtype=torch.float32)[:50])
pl.prep.run()
st = SamplerState(pl, 4); st.x.copy_(inp["x_T"].cuda()); st.coefs.fill_(0.5)
ctx = pl.ctx
ctx.lib.upk_kernel_launches(ctx.h, 1)
pl.body.run()
nk = ctx.lib.upk_kernel_launches(ctx.h, 1)
st.launch(False); torch.cuda.synchronize()
best = 1e9
for rep in range(4):
    pl.step.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40): st.launch(False)
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
v = os.environ["AB_VAR"]
print("%s=%s  forward %.3f ms  kernels/forward %d" % (v, os.environ[v], best, nk))
PY
done

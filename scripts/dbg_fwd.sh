cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | tail -25
import contextlib, io, os, sys, time, torch
sys.path.insert(0, os.getcwd())
import upgpt_amd
from upgpt_amd import synth
from upgpt_amd.engine import SamplerState
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
unet = model.model.diffusion_model
pl = unet.plan(8, 32, 32, 87, 50, "sampler")
pl.body.run()
torch.cuda.synchronize()
print("ok", len(pl.body.ops))
PY

"""Replays the UNet forward graph (B=8, latent HxW) N times — target for rocprofv3 --pmc passes."""
import contextlib, io, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd
from upgpt_amd import synth
from upgpt_amd.engine import SamplerState
H = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 5
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
unet = model.model.diffusion_model
inp = synth.synth_inputs(8, (H, W), 4, 87, 768, seed=0, text_only=True)
pl = unet.plan(8, H, W, 87, 50, "sampler")
pl.load_x_nchw(inp["x_T"].cuda(), 0, 0); pl.load_x_nchw(inp["c_concat"].cuda(), 4, pl.cin_pad)
pl.load_context(inp["c_crossattn"].cuda()); pl.t_rows.copy_(torch.arange(981, 0, -20, dtype=torch.float32)[:50])
pl.prep.run()
st = SamplerState(pl, 4); st.x.copy_(inp["x_T"].cuda()); st.coefs.fill_(0.5)
torch.cuda.synchronize()
for _ in range(N):
    st.launch(False)
torch.cuda.synchronize()
print("replayed", N, "forwards; igemm launches per forward:", sum(1 for c in pl.body.cls if c.startswith("igemm")))

"""VAE decode time (B=8, latent 32x32) under graph replay (dev tool)."""
import contextlib, io, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd
from upgpt_amd import synth
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
vp = model.first_stage_model._decode_plan(8, 32, 32, 0.18215)
z = torch.randn(8, 4, 32, 32)
for _ in range(3): vp.run(z)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): vp.run(z)
torch.cuda.synchronize(); print("UPGPT_VAE_GN_BYPRODUCT=%s vae decode %.3f ms, ops %d" % (os.environ.get("UPGPT_VAE_GN_BYPRODUCT", "1"), (time.perf_counter() - t0) / 10 * 1e3, len(vp.prog.ops)))

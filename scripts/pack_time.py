"""How long does packing the UNet / VAE weights take (once per weight set)? Dev tool."""
import contextlib, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd
from upgpt_amd import synth
with contextlib.redirect_stdout(io.StringIO()):
    m = upgpt_amd.build_model("bbox")
synth.fill_module_(m); m = m.cuda(); torch.cuda.synchronize()
t0 = time.perf_counter(); m.model.diffusion_model.packed(); torch.cuda.synchronize()
print("UNet pack (425 M params, incl. LayerNorm-folded variants): %.2f s" % (time.perf_counter() - t0))
t0 = time.perf_counter()
with m.ema_scope():
    m.model.diffusion_model.packed(); torch.cuda.synchronize()
print("UNet pack of the EMA weights: %.2f s" % (time.perf_counter() - t0))
t0 = time.perf_counter(); m.first_stage_model._decode_plan(8, 32, 32, 0.18215); torch.cuda.synchronize()
print("VAE decoder pack + plan B=8 32x32: %.2f s" % (time.perf_counter() - t0))
t0 = time.perf_counter(); m.model.diffusion_model.plan(8, 32, 32, 87, 50, "sampler"); torch.cuda.synchronize()
print("UNet plan B=8 32x32 (buffers + %d launch descriptors): %.2f s" % (0, time.perf_counter() - t0))

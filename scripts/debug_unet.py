"""GPU debug: product UNet forward vs the CPU oracle, per block (dev tool)."""
import sys, time
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd
from upgpt_amd import synth
from oracle import unet as o_unet

kind = sys.argv[1] if len(sys.argv) > 1 else "tiny"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfgs = {"tiny": synth.TINY_UNET, "bbox": synth.BBOX_UNET, "upscale": synth.UPSCALE_UNET}
model = upgpt_amd.build_model(kind)
sd = synth.fill_module_(model)
C = model.channels
ntok = 86 if kind == "upscale" else 87
inp = synth.synth_inputs(B, (32, 24), C, ntok, 768, seed=0, concat_channels=3 if kind == "upscale" else 1)
t = torch.tensor([981, 401, 21, 1, 500, 777, 901, 141][:B], dtype=torch.long)
x = torch.cat([inp["x_T"], inp["c_concat"]], 1)
taps = {}
t0 = time.time()
ref = o_unet.unet_forward(sd, cfgs[kind], x, t, inp["c_crossattn"], taps=taps)
print("oracle %.2fs" % (time.time() - t0))
model = model.cuda()
unet = model.model.diffusion_model
t0 = time.time()
out = unet(x.cuda(), t.cuda(), context=inp["c_crossattn"].cuda())
torch.cuda.synchronize()
print("hip first call %.2fs" % (time.time() - t0))
pl = unet.plan(B, 32, 24, ntok, B, "forward")
for name, act in pl.taps.items():
    got = act.t[:, :act.C].float().reshape(act.B, act.H, act.W, act.C).permute(0, 3, 1, 2).cpu()
    r = taps[name]
    err = (got - r).abs().max().item()
    print("%-18s shape %-22s max|ref| %8.3f  max err %9.5f  rel %.2e" % (name, tuple(r.shape), r.abs().max(), err, err / r.abs().max()))
err = (out.cpu() - ref).abs().max().item()
print("eps: max|ref| %.4f max err %.5f mse %.3e" % (ref.abs().max(), err, ((out.cpu() - ref) ** 2).mean()))
t0 = time.time()
for _ in range(5):
    out = unet(x.cuda(), t.cuda(), context=inp["c_crossattn"].cuda())
torch.cuda.synchronize()
print("hip eager fwd %.1f ms (launches: prep %d body %d)" % ((time.time() - t0) / 5 * 1e3, pl.prep.n_launch, pl.body.n_launch))

"""Dev probe: pins a halo-patch configuration on the 3x3 shapes whose key contains one of PROBE_KEYS (comma separated) and
prints forward time / kernels per forward before and after (input GroupNorm taken by the conv: fewer kernels)."""
import contextlib, ctypes as C, io, os, sys, time, traceback, torch
sys.path.insert(0, os.getcwd())
import upgpt_amd
from upgpt_amd import synth
from upgpt_amd.engine import SamplerState
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
unet = model.model.diffusion_model
B, HW = 8, (32, 32)
inp = synth.synth_inputs(B, HW, 4, 87, 768, seed=0, text_only=True)
pl = unet.plan(B, HW[0], HW[1], 87, 50, "sampler")
pl.load_x_nchw(inp["x_T"].cuda(), 0, 0); pl.load_x_nchw(inp["c_concat"].cuda(), 4, pl.cin_pad)
pl.load_context(inp["c_crossattn"].cuda()); pl.t_rows.copy_(torch.arange(981, 0, -20, dtype=torch.float32)[:50])
pl.prep.run()
st = SamplerState(pl, 4); st.x.copy_(inp["x_T"].cuda()); st.coefs.fill_(0.5)
ctx = pl.ctx
names = [ctx.lib.upk_conv_config_name(i).decode() for i in range(ctx.lib.upk_conv_num_configs())]


def measure(tag):
    for g in list(st.graphs.values()):
        ctx.graph_destroy(g)
    st.graphs.clear()
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
    print("%-60s forward %.4f ms  kernels %d" % (tag, best, nk), flush=True)
    return best


measure("tuned")
out0 = pl.eps.clone() if hasattr(pl, "eps") else None
body = set(id(k) for k in pl.body.keep)
subs = [s for s in os.environ.get("PROBE_KEYS", "M8192_N224_C224+0_k3s1").split(",") if s]
cfgs = [c for c in os.environ.get("PROBE_CFGS", "hc4x7p4,hc4x4p8").split(",") if c]
for sub in subs:
    ds = [(d, k) for d, k in pl.convs if id(d) in body and sub in k and d.ksize == 3]
    keys = sorted(set(k for _, k in ds))
    old = [(d.tune_cfg, d.tune_splitk) for d, _ in ds]
    for cn in cfgs:
        ci = names.index(cn)
        for d, _ in ds:
            d.tune_cfg, d.tune_splitk = ci + 1, 1
        try:
            measure("%s x%d %s -> %s" % (sub, len(ds), keys, cn))
        except Exception:
            traceback.print_exc()
    for (d, _), o in zip(ds, old):
        d.tune_cfg, d.tune_splitk = o
measure("tuned again")

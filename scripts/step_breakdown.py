"""Where one bench step (sample + decode) spends its time outside the 50 replayed forwards (dev tool)."""
import contextlib, io, os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import upgpt_amd
from upgpt_amd import synth
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
wl = bench.Workload(model, 8, (32, 32), 50, seed=0)
for _ in range(2):
    bench.quiet(wl.run)
torch.cuda.synchronize()
def t(fn, n=3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out
ms_step, _ = t(lambda: bench.quiet(wl.run))
def sample_only():
    with model.ema_scope():
        return wl.sampler.sample(wl.S, wl.B, (4,) + tuple(wl.hw), wl.cond, eta=0.0, x_T=wl.x_T, verbose=False, log_every_t=10 ** 6)[0]
ms_sample, z = t(lambda: bench.quiet(sample_only))
ms_dec, _ = t(lambda: model.decode_first_stage(z))
print("step %.2f ms = sample %.2f + decode %.2f (+ %.2f)" % (ms_step, ms_sample, ms_dec, ms_step - ms_sample - ms_dec))
pr = cProfile.Profile(); pr.enable(); bench.quiet(wl.run); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])

"""Dev tool: where the host time of one bench step (DDIM sample + VAE decode, B = 8, 32x32) goes outside the 50 graph
launches: cProfile of a warm Workload.run() (cumulative, top entries) + wall clock vs 50 x forward + decode."""
import contextlib, cProfile, io, os, pstats, sys, time, torch
sys.path.insert(0, os.getcwd())
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
ts = []
for _ in range(5):
    t0 = time.perf_counter(); bench.quiet(wl.run); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("step wall ms:", ["%.2f" % t for t in ts])
pr = cProfile.Profile()
pr.enable(); bench.quiet(wl.run); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])

"""Host-side cost of one bench step (sample() + decode enqueue), device idle at entry: cProfile, top entries."""
import contextlib, cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, upgpt_amd
from upgpt_amd import synth
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
wl = bench.Workload(model, 8, (32, 32), 50, seed=0)
with contextlib.redirect_stdout(io.StringIO()):
    wl.run(); wl.run()
torch.cuda.synchronize()
pr = cProfile.Profile()
with contextlib.redirect_stdout(io.StringIO()):
    t0 = time.perf_counter()
    pr.enable(); out = wl.run(); pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print("host enqueue %.2f ms, then device drain %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(45); print(st.getvalue()[:9000])

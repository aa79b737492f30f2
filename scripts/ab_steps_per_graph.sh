#!/bin/bash
# A/B on one box: wall time of one bench step (50-step DDIM + VAE decode, B = 8, 32x32) per UPGPT_STEPS_PER_GRAPH value
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for v in "$@" "$@"; do
  UPGPT_STEPS_PER_GRAPH=$v python - <<'PY' 2>/dev/null | tail -1
import contextlib, io, os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench, upgpt_amd
from upgpt_amd import synth
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
wl = bench.Workload(model, 8, (32, 32), 50, seed=0)
for _ in range(2):
    bench.quiet(wl.run)
torch.cuda.synchronize()
ts = []
for _ in range(6):
    t0 = time.perf_counter(); bench.quiet(wl.run); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("UPGPT_STEPS_PER_GRAPH=%s  step %.2f ms (min of 6; median %.2f)  %.2f img/s" % (os.environ["UPGPT_STEPS_PER_GRAPH"], min(ts), sorted(ts)[3], 8e3 / min(ts)))
PY
done | tee gpurun_out/ab_steps.log

"""Experiment: K bench steps (50-step DDIM sample + VAE decode, B = 8, 32x32) with the decode of batch k on a second stream
while batch k + 1 is being sampled, against the serial loop.  Same work, same results; only the order on the device changes."""
import contextlib, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import upgpt_amd
from upgpt_amd import synth
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
wl = bench.Workload(model, 8, (32, 32), 50, seed=0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
side = torch.cuda.Stream()

def sample():
    with model.ema_scope():
        z, _ = wl.sampler.sample(wl.S, wl.B, (4,) + tuple(wl.hw), wl.cond, eta=0.0, x_T=wl.x_T, verbose=False, log_every_t=10 ** 6)
    return z

def serial():
    out = None
    for _ in range(K):
        out = model.decode_first_stage(bench.quiet(sample))
    return out

def overlapped():
    out = None
    main = torch.cuda.current_stream()
    for _ in range(K):
        z = bench.quiet(sample)
        ev = torch.cuda.Event(); ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            z.record_stream(side)
            out = model.decode_first_stage(z)
    main.wait_stream(side)
    return out

for name, fn in (("serial", serial), ("overlapped", overlapped), ("serial", serial), ("overlapped", overlapped)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-10s %d steps: %.2f ms per step = %.2f img/s   (checksum %.6f)" % (name, K, dt / K * 1e3, 8 * K / dt, float(out.double().sum())), flush=True)

"""Measures the per-kernel floor: N dependent trivial kernels, eager vs HIP-graph replay."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upgpt_amd._lib import get_context
ctx = get_context(0)
step = torch.zeros(1, dtype=torch.int32, device="cuda")
x = torch.randn(8192, 224, device="cuda").half(); y = torch.empty_like(x)
g = torch.ones(224, device="cuda"); b = torch.zeros(224, device="cuda")
N = 400
def run_adv():
    for _ in range(N): ctx.advance_step(step)
def run_ln():
    for _ in range(N): ctx.layernorm(x, 224, 8192, 224, g, b, 1e-5, y, 224)
for name, fn in (("advance_step (1 thread)", run_adv), ("layernorm 8192x224", run_ln)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); te = (time.perf_counter() - t0) / N * 1e6
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx.graph_begin(); fn(); gr = ctx.graph_end()
        ctx.graph_launch(gr); s.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): ctx.graph_launch(gr)
        s.synchronize(); tg = (time.perf_counter() - t0) / 5 / N * 1e6
    print("%-26s eager %.2f us/kernel   graph %.2f us/kernel" % (name, te, tg))

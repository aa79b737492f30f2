"""Runs ONE conv/GEMM shape N times under HIP-graph replay (for rocprofv3 --pmc). Dev tool.
   python scripts/one_conv.py B H W cin cout ks cfg sk [reps]"""
import sys, math, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upgpt_amd import _lib as L
B, H, W, cin, cout, ks, cfg, sk = (int(v) for v in sys.argv[1:9])
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 20
ctx = L.get_context(0)
x = torch.randn(B * H * W, cin, device="cuda").half()
w = (torch.randn(cout, cin, ks, ks, device="cuda") / math.sqrt(cin * ks * ks)).contiguous()
if os.environ.get("ZERO") == "1":  # all-zero operands: the MFMA rate without the data-dependent switching power
    x.zero_(); w.zero_()
wp, n_pad = ctx.pack_weight(w)
y = torch.empty(B * H * W, cout, device="cuda", dtype=torch.float16)
d = L.ConvDesc()
d.x1 = x.data_ptr(); d.c1 = cin; d.ld1 = cin; d.batch = B; d.in_h = H; d.in_w = W; d.ksize = ks; d.stride = 1
d.w_packed = wp.data_ptr(); d.n_out = cout; d.n_pad = n_pad; d.y = y.data_ptr(); d.ldy = cout
d.tune_cfg = cfg + 1; d.tune_splitk = sk
epi = os.environ.get("EPI", "")  # e.g. EPI=bias,res
if "bias" in epi:
    bias = torch.randn(n_pad, device="cuda"); d.bias = bias.data_ptr()
if "geglu" in epi:
    d.flags = 2; d.n_out = cout // 2; d.ldy = cout // 2
if "res" in epi:
    res = torch.randn(B * H * W, cout, device="cuda").half(); d.residual = res.data_ptr(); d.ld_res = cout
ctx.conv(d); torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx.graph_begin()
    for _ in range(reps): ctx.conv(d)
    g = ctx.graph_end()
    ctx.graph_launch(g); s.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(5): ctx.graph_launch(g)
    e1.record(s); s.synchronize()
us = e0.elapsed_time(e1) / 5 / reps * 1e3
gf = 2 * B * H * W * cout * cin * ks * ks / 1e9
print("shape B%d %dx%d %d->%d k%d cfg %s sk %d: %.2f us/launch (graph) = %.0f TF/s" % (
    B, H, W, cin, cout, ks, ctx.lib.upk_conv_config_name(cfg).decode(), sk, us, gf / us * 1e-3 * 1e3 / 1e3 * 1e3))

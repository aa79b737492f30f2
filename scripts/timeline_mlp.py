"""In-kernel timeline of the fused feed-forward launch (dev library built with -DUPK_TIMELINE, UPK_MLP_TL=1).
   python scripts/timeline_mlp.py M C rows"""
import os, sys, math, ctypes as C
os.environ["UPK_MLP_TL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upgpt_amd import _lib as L
M, c, rows = (int(v) for v in sys.argv[1:4])
inner = 4 * c
ctx = L.get_context(0)
x = torch.randn(M, c, device="cuda").half(); res = torch.randn(M, c, device="cuda").half()
w1p, n1 = ctx.pack_weight((torch.randn(2 * inner, c, device="cuda") / math.sqrt(c)).contiguous())
w2a, n_pad = ctx.pack_weight((torch.randn(c, inner, device="cuda") / math.sqrt(inner)).contiguous())
w2b, _ = ctx.pack_weight((torch.randn(c, c, device="cuda") / math.sqrt(c)).contiguous())
w2p = torch.cat([w2a.reshape(-1), w2b.reshape(-1)]).contiguous()
b1 = torch.zeros(n1, device="cuda"); u1 = torch.zeros(n1, device="cuda"); b2 = torch.zeros(n_pad, device="cuda")
y = torch.zeros(M, c, device="cuda", dtype=torch.float16)
d = L.MlpDesc()
d.x, d.ldx, d.m, d.c, d.inner = x.data_ptr(), c, M, c, inner
d.w1, d.b1, d.u1, d.ln_eps, d.ln_dim = w1p.data_ptr(), b1.data_ptr(), u1.data_ptr(), 1e-5, c
d.w2, d.b2, d.n_out, d.n_pad = w2p.data_ptr(), b2.data_ptr(), c, n_pad
d.residual, d.ld_res, d.y, d.ldy, d.rows_per_wg = res.data_ptr(), c, y.data_ptr(), c, rows
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
ws = ctx.workspace
names = {1: "issued", 2: "bar", 3: "ln", 20: "s1end", 21: "bar2", 22: "K2", 23: "drained"}
for p in range(8):
    names[4 + 2 * p] = "K%d" % p; names[5 + 2 * p] = "E%d" % p
for trial in range(4):
    if trial >= 2: flush.fill_(trial)
    ws[-4096:].zero_(); torch.cuda.synchronize()
    ctx._chk(ctx.lib.upk_geglu_mlp_f16(ctx.h, C.byref(d), ctx._s())); torch.cuda.synchronize()
    st = ws[-4096:].view(torch.int64).cpu().numpy()
    for blk, off in (("first w0", 0), ("first w4", 32), ("last w0", 64)):
        t = [int(st[off + k]) for k in range(32)]
        if not t[0]: continue
        print("trial %d (%s) %-8s: " % (trial, "flushed" if trial >= 2 else "warm", blk) +
              "  ".join("%s +%d" % (names.get(k, str(k)), t[k] - t[0]) for k in sorted(range(1, 32), key=lambda k: t[k]) if t[k]))

"""In-kernel timeline of the fused cross-attention launch (dev library built with -DUPK_TIMELINE, UPK_XB_TL=1).
   python scripts/timeline_xb.py B hw C dp rows"""
import os, sys, math, ctypes as C
os.environ["UPK_XB_TL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upgpt_amd import _lib as L
B, hw, c, dp, rows = (int(v) for v in sys.argv[1:6])
M, hd, nkv = B * hw, 8 * dp, 87
ctx = L.get_context(0)
dev = "cuda"
a1 = torch.randn(M, hd, device=dev).half(); t0 = torch.randn(M, c, device=dev).half()
w1p, _ = ctx.pack_weight((torch.randn(c, hd, device=dev) / math.sqrt(hd)).contiguous())
w3p, _ = ctx.pack_weight((torch.randn(c, hd, device=dev) / math.sqrt(hd)).contiguous())
wqp, _ = ctx.pack_weight((torch.randn(hd, c, device=dev) / math.sqrt(c)).contiguous())
vec = torch.zeros(((2 * c + 2 * hd + 255) // 256) * 256, device=dev)
kc = torch.randn(B * nkv, hd, device=dev).half(); vt = torch.randn(B, 8, dp, 96, device=dev).half()
y = torch.zeros(M, c, device=dev, dtype=torch.float16)
d = L.XblockDesc()
d.a1, d.lda, d.m, d.c, d.heads, d.d = a1.data_ptr(), hd, M, c, 8, dp
d.t0, d.ld_t0 = t0.data_ptr(), c
d.w_out1, d.w_q, d.w_out2, d.vec = w1p.data_ptr(), wqp.data_ptr(), w3p.data_ptr(), vec.data_ptr()
d.ln_eps, d.ln_dim = 1e-5, c
d.k_ctx, d.ldk, d.n_kv = kc.data_ptr(), hd, nkv
d.vt_ctx, d.vt_ld, d.scale = vt.data_ptr(), 96, 0.19
d.y, d.ldy, d.hw, d.rows_per_wg = y.data_ptr(), c, hw, rows
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
ws = ctx.workspace
names = {1: "issued", 2: "bar0", 3: "G1", 4: "bar1", 5: "LN", 6: "G2", 7: "XA", 8: "bar2", 9: "G3", 10: "drained"}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for trial in range(4):
    if trial >= 2: flush.fill_(trial)
    ws[-4096:].zero_(); torch.cuda.synchronize()
    e0.record(); ctx._chk(ctx.lib.upk_cross_block_f16(ctx.h, C.byref(d), ctx._s())); e1.record(); torch.cuda.synchronize()
    st = ws[-4096:].view(torch.int64).cpu().numpy()
    for blk, off in (("first w0", 0), ("first w4", 32), ("last w0", 64)):
        t = [int(st[off + k]) for k in range(32)]
        if not t[0]: continue
        print("trial %d (%s, %.1f us) %-8s: " % (trial, "flushed" if trial >= 2 else "warm", e0.elapsed_time(e1) * 1e3, blk) +
              "  ".join("%s +%d" % (names.get(k, str(k)), t[k] - t[0]) for k in sorted(range(1, 32), key=lambda k: t[k]) if t[k]))

"""In-kernel timeline of one WS igemm launch (UPK_ABLATE=0x200000 stamps; dev tool).
   python scripts/timeline.py B H W cin cout ks cfg sk"""
import os, sys, math, ctypes
os.environ["UPK_CXXFLAGS"] = "-DUPK_TIMELINE"  # needs a stamp-enabled build: rm upgpt_amd/libupk.so first
os.environ["UPK_ABLATE"] = hex(int(os.environ.get("UPK_ABLATE", "0"), 0) | 0x200000)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upgpt_amd import _lib as L
B, H, W, cin, cout, ks, cfg, sk = (int(v) for v in sys.argv[1:9])
ctx = L.get_context(0)
x = torch.randn(B * H * W, cin, device="cuda").half()
w = (torch.randn(cout, cin, ks, ks, device="cuda") / math.sqrt(cin * ks * ks)).contiguous()
wp, n_pad = ctx.pack_weight(w)
y = torch.empty(B * H * W, cout, device="cuda", dtype=torch.float16)
bias = torch.randn(n_pad, device="cuda")
res = torch.randn(B * H * W, cout, device="cuda").half()
d = L.ConvDesc()
d.x1 = x.data_ptr(); d.c1 = cin; d.ld1 = cin; d.batch = B; d.in_h = H; d.in_w = W; d.ksize = ks; d.stride = 1
d.w_packed = wp.data_ptr(); d.n_out = cout; d.n_pad = n_pad; d.y = y.data_ptr(); d.ldy = cout
if os.environ.get('NOEPIOPS', '0') != '1':
    d.bias = bias.data_ptr(); d.residual = res.data_ptr(); d.ld_res = cout
d.tune_cfg = cfg + 1; d.tune_splitk = sk
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
ws = ctx.workspace  # torch uint8 tensor
names = {0: "c.entry", 1: "c.stage0 ready", 2: "c.stage1 start", 3: "c.loop end", 4: "c.stores done",

         8: "l.entry", 9: "l.setup done", 10: "l.prologue issued", 11: "l.stage0 landed", 12: "l.loop end"}
for trial in range(4):
    if trial >= 2:
        flush.fill_(trial)  # evict L2 / Infinity Cache
    ws[-4096:].zero_()
    torch.cuda.synchronize()
    ctx.conv(d); torch.cuda.synchronize()
    st = ws[-4096:].view(torch.int64).cpu().numpy()
    for blk, off in (("first", 0), ("last", 32)):
        t = {k: int(st[off + k]) for k in names if st[off + k]}
        if not t: continue
        t0 = min(t.values())
        print("trial %d (%s) block %-5s: " % (trial, "flushed" if trial >= 2 else "warm", blk) +
              "  ".join("%s +%d" % (names[k], t[k] - t0) for k in sorted(t, key=lambda k: t[k])))

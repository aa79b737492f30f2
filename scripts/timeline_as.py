"""In-kernel timeline of one A-stationary launch (dev library built with -DUPK_TIMELINE; stamps of waves 0 and 4 of the
first and the last workgroup).   python scripts/timeline_as.py M K N cfgname ppw [geglu]"""
import os, sys, math
os.environ["UPK_ABLATE"] = hex(int(os.environ.get("UPK_ABLATE", "0"), 0) | 0x200000)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upgpt_amd import _lib as L
M, K, N = (int(v) for v in sys.argv[1:4])
cfgname, ppw = sys.argv[4], int(sys.argv[5])
geglu = len(sys.argv) > 6 and sys.argv[6] == "geglu"
ctx = L.get_context(0)
cfg = [i for i in range(ctx.lib.upk_conv_num_configs()) if ctx.lib.upk_conv_config_name(i).decode() == cfgname][0]
x = torch.randn(M, K, device="cuda").half()
w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).contiguous()
wp, n_pad = ctx.pack_weight(w)
n_out = N // 2 if geglu else N
y = torch.empty(M, n_out, device="cuda", dtype=torch.float16)
bias = torch.randn(n_pad, device="cuda")
d = L.ConvDesc()
d.x1 = x.data_ptr(); d.c1 = K; d.ld1 = K; d.batch = 1; d.in_h = M; d.in_w = 1; d.ksize = 1; d.stride = 1
d.w_packed = wp.data_ptr(); d.n_out = n_out; d.n_pad = n_pad; d.y = y.data_ptr(); d.ldy = n_out; d.bias = bias.data_ptr()
d.flags = L.F_GEGLU if geglu else 0
d.tune_cfg = cfg + 1; d.tune_splitk = ppw
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
ws = ctx.workspace
for trial in range(4):
    if trial >= 2:
        flush.fill_(trial)
    ws[-4096:].zero_()
    torch.cuda.synchronize()
    ctx.conv(d); torch.cuda.synchronize()
    st = ws[-4096:].view(torch.int64).cpu().numpy()
    for blk, off in (("first w0", 0), ("first w4", 32), ("last w0", 64), ("last w4", 96)):
        t = [int(st[off + k]) for k in range(32)]
        if not t[0]:
            continue
        t0 = t[0]
        names = {1: "dma", 2: "ring", 3: "bar", 4: "ln", 31: "drained"}
        names.update({16 + u: "s%d" % u for u in range(8)})
        out = []
        for k in sorted(range(1, 32), key=lambda k: t[k]):
            if t[k]:
                nm = names.get(k, ("K%d" % ((k - 5) // 2)) if (k - 5) % 2 == 0 else ("E%d" % ((k - 6) // 2)))
                out.append("%s +%d" % (nm, t[k] - t0))
        print("trial %d (%s) %-8s: " % (trial, "flushed" if trial >= 2 else "warm", blk) + "  ".join(out))

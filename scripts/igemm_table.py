"""Per-(tile config, split-K) timing table for representative UNet conv/GEMM shapes (dev tool)."""
import sys, math, ctypes as C
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upgpt_amd import _lib as L
ctx = L.get_context(0)
lib = ctx.lib
dev = "cuda"
SHAPES = [  # name, B, H, W, cin, cout, ks, flags
    ("conv3 224->224 M8192", 8, 32, 32, 224, 224, 3, 0),
    ("conv3 448->448 M2048", 8, 16, 16, 448, 448, 3, 0),
    ("conv3 896->896 M512", 8, 8, 8, 896, 896, 3, 0),
    ("conv3 896->896 M128", 8, 4, 4, 896, 896, 3, 0),
    ("conv3 1792->896 M512", 8, 8, 8, 1792, 896, 3, 0),
    ("gemm 224->224 M8192", 1, 8192, 1, 224, 224, 1, 0),
    ("gemm 224->768 M8192", 1, 8192, 1, 224, 768, 1, 0),
    ("gemm 896->224 M8192", 1, 8192, 1, 896, 224, 1, 0),
    ("gemm 448->448 M2048", 1, 2048, 1, 448, 448, 1, 0),
    ("gemm 896->896 M512", 1, 512, 1, 896, 896, 1, 0),
    ("gemm 3584->896 M512", 1, 512, 1, 3584, 896, 1, 0),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
ncfg = lib.upk_conv_num_configs()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, B, H, W, cin, cout, ks, flags in SHAPES:
    if only and only not in name: continue
    x = torch.randn(B * H * W, cin, device=dev).half()
    w = (torch.randn(cout, cin, ks, ks, device=dev) / math.sqrt(cin * ks * ks)).contiguous()
    wp, n_pad = ctx.pack_weight(w)
    y = torch.empty(B * H * W, cout, device=dev, dtype=torch.float16)
    d = L.ConvDesc()
    d.x1 = x.data_ptr(); d.c1 = cin; d.ld1 = cin; d.batch = B; d.in_h = H; d.in_w = W; d.ksize = ks; d.stride = 1
    d.w_packed = wp.data_ptr(); d.n_out = cout; d.n_pad = n_pad; d.y = y.data_ptr(); d.ldy = cout
    M = B * H * W
    gf = 2 * M * cout * cin * ks * ks / 1e9
    res = []
    for cfg in range(ncfg):
        for sk in (1, 2, 3, 4, 6, 8, 9, 12, 16, 18):
            ctx.conv_override(cfg, sk)
            try:
                ctx.conv(d)
            except L.UpkError:
                continue
            e0.record()
            for _ in range(10): ctx.conv(d)
            e1.record(); torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1) / 10 * 1e3, cfg, sk))
    ctx.conv_override(-1, 0)
    res.sort()
    print("== %s  %.2f GF" % (name, gf))
    for us, cfg, sk in res[:6]:
        print("   %7.1f us  %6.0f TF/s  cfg %-9s sk %d" % (us, gf / us * 1e3, lib.upk_conv_config_name(cfg).decode(), sk))  # GF / us = 1e3 TF/s
    nosplit = [r for r in res if r[2] == 1][:3]
    for us, cfg, sk in nosplit:
        print("   (no split) %7.1f us  %6.0f TF/s  cfg %s" % (us, gf / us * 1e3, lib.upk_conv_config_name(cfg).decode()))

"""Times the patch kernel (csrc/pconv.hip) per tile configuration on the UNet's conv shapes, with the weights
cycling through enough copies that they come from HBM as inside the forward (dev tool).

  python scripts/pc_bench.py [shape-name ...]      (UPK_PC_CFGS=0,3 restricts the configurations)
"""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upgpt_amd import _lib as L
from upgpt_amd._lib import get_context

DEV = "cuda"
# name: (ks, c1, c2, c_app, cout, H, W, gn)
SHAPES = {
    "L0a": (3, 224, 0, 0, 224, 32, 32, 1), "L0c": (3, 448, 224, 0, 224, 32, 32, 1), "L0k": (3, 224, 0, 672, 224, 32, 32, 1),
    "L1a": (3, 448, 0, 0, 448, 16, 16, 1), "L1c": (3, 896, 448, 0, 448, 16, 16, 1), "L1k": (3, 448, 0, 1344, 448, 16, 16, 1),
    "L2a": (3, 896, 0, 0, 896, 8, 8, 1), "L2c": (3, 896, 896, 0, 896, 8, 8, 1), "L2k": (3, 896, 0, 1792, 896, 8, 8, 1),
    "L3a": (3, 896, 0, 0, 896, 4, 4, 1), "L3c": (3, 896, 896, 0, 896, 4, 4, 1),
    "V3a": (3, 128, 0, 0, 128, 256, 256, 1), "V2a": (3, 256, 0, 0, 256, 128, 128, 1), "V1a": (3, 512, 0, 0, 512, 64, 64, 1),
    "V0a": (3, 512, 0, 0, 512, 32, 32, 1),
    "P0": (1, 224, 0, 0, 224, 32, 32, 1), "P2": (1, 896, 0, 0, 896, 8, 8, 1), "L0n": (3, 224, 0, 0, 224, 32, 32, 0),
}


def main():
    ctx = get_context(0)
    names = [a for a in sys.argv[1:] if a in SHAPES] or list(SHAPES)
    only = [int(v) for v in os.environ.get("UPK_PC_CFGS", "").split(",") if v]
    B = 8
    ncfg = ctx.lib.upk_pconv_num_configs()
    for name in names:
        ks, c1, c2, ca, cout, H, W, gn = SHAPES[name]
        reps = 20 if H * W <= 1024 else 4
        g = torch.Generator().manual_seed(1)
        x1 = torch.randn(B, H, W, c1, generator=g).half().to(DEV)
        x2 = torch.randn(B, H, W, c2, generator=g).half().to(DEV) if c2 else None
        x3 = torch.randn(B, H, W, ca, generator=g).half().to(DEV) if ca else None
        K = ks * ks * (c1 + c2) + ca
        wbytes = K * cout * 2
        ncopy = max(2, min(64, int(400e6 // wbytes)))
        w = (torch.randn(cout, c1 + c2, ks, ks, generator=g) / math.sqrt(K)).to(DEV)
        wp, n_pad = ctx.pack_weight(w.contiguous())
        if ca:
            wa, _ = ctx.pack_weight((torch.randn(cout, ca, 1, 1, generator=g) / math.sqrt(K)).to(DEV).contiguous())
            wp = torch.cat([wp.reshape(-1), wa.reshape(-1)])
        wps = [wp.clone() for _ in range(ncopy)]
        bias = torch.zeros(n_pad, device=DEV)
        y = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
        gamma, beta = torch.ones(c1 + c2, device=DEV), torch.zeros(c1 + c2, device=DEV)
        ws = torch.zeros(ctx.groupnorm_ws_bytes(B, H * W) // 4 + 64, device=DEV)
        ctx._chk(ctx.lib.upk_groupnorm_stats_nhwc_f16(ctx.h, x1.data_ptr(), c1, c1, x2.data_ptr() if c2 else None, c2, c2,
                                                      B, H * W, 32, ws.data_ptr(), ctx._s()))
        sws = torch.zeros(ctx.gn_stats_floats(B, n_pad), device=DEV)
        gf = 2.0 * B * H * W * cout * K / 1e9
        print("== %s: k%d %d+%d(+%d) -> %d @%dx%d  %.2f GF, weights %.1f MB x %d copies" % (
            name, ks, c1, c2, ca, cout, H, W, gf, wbytes / 1e6, ncopy), flush=True)
        res = []
        for cfg in range(ncfg):
            if only and cfg not in only:
                continue
            d = L.ConvDesc()
            d.x1, d.c1, d.ld1 = x1.data_ptr(), c1, c1
            if c2:
                d.x2, d.c2, d.ld2 = x2.data_ptr(), c2, c2
            if ca:
                d.x3, d.c3, d.ld3 = x3.data_ptr(), ca, ca
            d.batch, d.in_h, d.in_w, d.ksize, d.stride = B, H, W, ks, 1
            d.w_packed, d.n_out, d.n_pad, d.bias = wps[0].data_ptr(), cout, n_pad, bias.data_ptr()
            d.y, d.ldy = y.data_ptr(), cout
            d.pc_enable, d.pc_cfg = 1, cfg + 1
            d.gn_stats_ws, d.gn_groups = sws.data_ptr(), 32
            if gn:
                d.gni_mode, d.gni_silu, d.gni_groups, d.gni_eps = 1, 1, 32, 1e-5
            if not ctx.lib.upk_pconv_supported(ctx.h, C.byref(d)):
                continue
            if gn:
                d.gni_gamma, d.gni_beta = gamma.data_ptr(), beta.data_ptr()
                d.gni_stats1, d.gni_nblk1 = ws.data_ptr(), ctx.lib.upk_groupnorm_chunks(H * W)
            ctx.conv(d)
            torch.cuda.synchronize()
            if os.environ.get("UPK_PC_STAMPS"):  # dev library + UPK_ABLATE bit 0x200000: in-kernel s_memtime stamps
                names = ["entry", "ring issued", "x issued", "stats done", "staged", "ring landed", "barrier", "slab0 done",
                         "loop done", "reduced", "stored"]
                for trial in range(3):
                    ctx.workspace[-4096:].zero_()
                    d.w_packed = wps[(trial + 1) % ncopy].data_ptr()
                    ctx.conv(d)
                    torch.cuda.synchronize()
                    stv = ctx.workspace[-4096:].view(torch.int64).cpu().numpy()
                    for blk, off in (("first", 0), ("last", 32)):
                        for wv, o2 in (("mfma", 0), ("load", 16)):
                            t = {k: int(stv[off + o2 + k]) for k in range(11) if stv[off + o2 + k]}
                            if t:
                                t0 = t.get(0, min(t.values()))
                                print("      [%s %s %s] " % (ctx.lib.upk_pconv_config_name(cfg).decode(), blk, wv) +
                                      "  ".join("%s +%d" % (names[k], t[k] - t0) for k in sorted(t)))
            # `reps` launches captured into one graph (eager launches from Python are host-bound at ~14 us each)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                ctx.graph_begin()
                for r in range(reps):
                    d.w_packed = wps[r % ncopy].data_ptr()
                    ctx.conv(d)
                gr = ctx.graph_end()
                ctx.graph_launch(gr)
                st.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    ctx.graph_launch(gr)
                e1.record()
                st.synchronize()
                ctx.graph_destroy(gr)
            us = e0.elapsed_time(e1) * 1e3 / reps / 3
            res.append((us, ctx.lib.upk_pconv_config_name(cfg).decode(), cfg))
        for us, nm, cfg in sorted(res):
            print("   %7.1f us  %6.0f TF/s  %5.2f TB/s(w)   cfg %2d %s" % (us, gf / us * 1e3, wbytes / us / 1e6, cfg, nm), flush=True)


if __name__ == "__main__":
    main()

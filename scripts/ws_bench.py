"""Weight-streaming check of the implicit-GEMM kernels on the deep UNet levels (dev tool): every (tile config, split-K)
under graph replay, with the weights (a) the same buffer every launch (Infinity-Cache warm) and (b) cycling through
>= 600 MB of copies (HBM cold, as inside the forward).

  python scripts/ws_bench.py [shape-name ...]
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
# name: (ks, cin, cout, H, W, flags)
SHAPES = {
    "ff1_M512": (1, 896, 7168, 8, 8, L.F_GEGLU), "qkv_M512": (1, 896, 3072, 8, 8, 0), "ff2_M512": (1, 3584, 896, 8, 8, 0),
    "c3_M512": (3, 896, 896, 8, 8, 0), "c3_M128": (3, 896, 896, 4, 4, 0), "k1_M512": (1, 896, 896, 8, 8, 0),
    "ff1_M2048": (1, 448, 3584, 16, 16, L.F_GEGLU), "c3_M2048": (3, 448, 448, 16, 16, 0),
    "ff1_M8192": (1, 224, 1792, 32, 32, L.F_GEGLU), "qkv_M8192": (1, 224, 768, 32, 32, 0), "ff2_M8192": (1, 896, 224, 32, 32, 0),
    "v128": (3, 128, 128, 256, 256, 0), "v256": (3, 256, 256, 128, 128, 0), "v512": (3, 512, 512, 64, 64, 0),
    "v512s": (3, 512, 512, 32, 32, 0), "v256_128": (3, 256, 128, 256, 256, 0),
    "qkv_M2048": (1, 448, 1536, 16, 16, 0), "ff2_M2048": (1, 1792, 448, 16, 16, 0), "ff2a_M8192": (1, 1120, 224, 32, 32, 0),
    "ff2a_M2048": (1, 2240, 448, 16, 16, 0), "to_out_M8192": (1, 256, 224, 32, 32, 0), "to_out_M2048": (1, 512, 448, 16, 16, 0),
    "ff1_M128": (1, 896, 7168, 4, 4, L.F_GEGLU), "qkv_M128": (1, 896, 3072, 4, 4, 0),
    "c3_M8192": (3, 224, 224, 32, 32, 0), "k1_M8192": (1, 224, 224, 32, 32, 0), "c3_M8192_448": (3, 448, 224, 32, 32, 0),
}


def main():
    ctx = get_context(0)
    names = [a for a in sys.argv[1:] if a.split("+")[0] in SHAPES] or list(SHAPES)  # name[+gs][+rv][+res]
    B = 8
    ncfg = ctx.lib.upk_conv_num_configs()
    only = [v.split(":") for v in os.environ.get("UPK_WS_ONLY", "").split(",") if v]  # e.g. as4x2p7:4,2x4x2x2k2w3:1
    for name in names:
        base = name.split("+")[0]
        reps = 16 if SHAPES[base][3] * SHAPES[base][4] <= 4096 else 3
        feats = name.split("+")[1:]
        ks, cin, cout, H, W, flags = SHAPES[base]
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, H, W, cin, generator=g).half().to(DEV)
        K = ks * ks * cin
        wbytes = K * cout * 2
        w = (torch.randn(cout, cin, ks, ks, generator=g) / math.sqrt(K)).to(DEV)
        wp, n_pad = ctx.pack_weight(w.contiguous())
        ncold = max(2, min(64, int(600e6 // wbytes)))
        wps = [wp.clone() for _ in range(ncold)]
        n_real = cout // 2 if flags & L.F_GEGLU else cout
        y = torch.zeros(B, H, W, n_real, device=DEV, dtype=torch.float16)
        bias = torch.zeros(n_pad, device=DEV)
        sws = torch.zeros(ctx.gn_stats_floats(B, n_pad), device=DEV)
        rowv = torch.zeros(50, n_pad, device=DEV)
        stepc = torch.zeros(1, dtype=torch.int32, device=DEV)
        resid = torch.zeros(B, H, W, n_real, device=DEV, dtype=torch.float16)
        gf = 2.0 * B * H * W * cout * K / 1e9
        print("== %s: k%d %d -> %d @%dx%d M=%d  %.2f GF, weights %.1f MB (cold: %d copies)" % (
            name, ks, cin, cout, H, W, B * H * W, gf, wbytes / 1e6, ncold), flush=True)
        res = {}
        for cfg in range(ncfg):
            cname = ctx.lib.upk_conv_config_name(cfg).decode()
            is_as = cname.startswith("as")
            if only and cname not in [o[0] for o in only]:
                continue
            if is_as and ks != 1:
                continue
            for sk in ((1, 2, 3, 4, 6, 8, 12, 16) if is_as else (1, 2, 4, 8, 9) if B * H * W <= 8192 else (1,)):
                if only and [cname, str(sk)] not in only:
                    continue
                d = L.ConvDesc()
                d.x1, d.c1, d.ld1 = x.data_ptr(), cin, cin
                d.batch, d.in_h, d.in_w, d.ksize, d.stride = B, H, W, ks, 1
                d.w_packed, d.n_out, d.n_pad, d.bias = wps[0].data_ptr(), cout, n_pad, bias.data_ptr()
                d.y, d.ldy, d.flags = y.data_ptr(), n_real, flags
                d.tune_cfg, d.tune_splitk = cfg + 1, sk
                if "gs" in feats:
                    d.gn_stats_ws, d.gn_groups = sws.data_ptr(), 32
                if "rv" in feats:
                    d.rowvec, d.rv_batch_stride, d.rv_step_stride, d.step = rowv.data_ptr(), 0, n_pad, stepc.data_ptr()
                if "res" in feats:
                    d.residual, d.ld_res = resid.data_ptr(), n_real
                try:
                    ctx.conv(d)
                except L.UpkError:
                    continue
                torch.cuda.synchronize()
                for mode, ncopy in (("warm", 1), ("cold", ncold)):
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
                        nl = 3 if ncopy == 1 else max(1, ncopy // reps)
                        for _ in range(nl):
                            ctx.graph_launch(gr)
                        e1.record()
                        st.synchronize()
                        ctx.graph_destroy(gr)
                    res.setdefault((cfg, sk), {})[mode] = e0.elapsed_time(e1) * 1e3 / reps / nl
        rows = sorted(res.items(), key=lambda kv: kv[1]["cold"])
        for (cfg, sk), t in rows[:8]:
            print("   cold %6.1f us (%5.2f TB/s w)  warm %6.1f us   cfg %-12s sk %d" % (
                t["cold"], wbytes / t["cold"] / 1e6, t["warm"], ctx.lib.upk_conv_config_name(cfg).decode(), sk), flush=True)
        bw = min(res.items(), key=lambda kv: kv[1]["warm"])
        print("   best warm: %.1f us  cfg %s sk %d (cold %.1f)" % (bw[1]["warm"], ctx.lib.upk_conv_config_name(bw[0][0]).decode(),
                                                                    bw[0][1], bw[1]["cold"]), flush=True)


if __name__ == "__main__":
    main()

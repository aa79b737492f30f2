"""Parity of the A-stationary patch kernel (csrc/pconv.hip: LDS-resident halo patch, GroupNorm + SiLU of the input
folded into its staging pass) against plain PyTorch fp32 references, through the C ABI."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from upgpt_amd import _lib as L
from test_ops_gpu import DEV, check, make_desc, nhwc16, rnd

pytestmark = pytest.mark.gpu


def pc_names(ctx):
    return [ctx.lib.upk_pconv_config_name(i).decode() for i in range(ctx.lib.upk_pconv_num_configs())]


def supported(ctx, d):
    return bool(ctx.lib.upk_pconv_supported(ctx.h, C.byref(d)))


SHAPES = [(3, 32, 32), (3, 16, 16), (3, 8, 8), (3, 4, 4), (3, 32, 24), (3, 16, 12), (3, 8, 6), (3, 4, 3), (3, 12, 10),
          (1, 32, 32), (1, 8, 6), (1, 4, 3)]


@pytest.mark.parametrize("ks,H,W", SHAPES)
def test_pconv_every_config_matches_conv2d(ctx, ks, H, W):
    """Plain conv (bias + residual + per-sample row vector) on every tile configuration that takes the shape."""
    B, cin, cout = 2, 96, 80
    x = rnd(B, cin, H, W)
    w = rnd(cout, cin, ks, ks, scale=1 / math.sqrt(ks * ks * cin))
    b = rnd(cout, scale=0.1)
    res = rnd(B, H, W, cout, seed=7).half()
    rv = rnd(B, cout, seed=8)
    ref = F.conv2d(x.half().float(), w.half().float(), b, padding=ks // 2) + res.float().permute(0, 3, 1, 2) \
        + rv.view(B, cout, 1, 1)
    xn = nhwc16(x)
    ran = 0
    for cfg, name in enumerate(pc_names(ctx)):
        y = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
        d = make_desc(ctx, xn, w, b, y, residual=res, rowvec=rv, rv_bs=cout)
        d.pc_enable, d.pc_cfg = 1, cfg + 1
        if not supported(ctx, d):
            continue
        ctx.conv(d)
        torch.cuda.synchronize()
        check(y.permute(0, 3, 1, 2), ref)
        ran += 1
    assert ran >= 2, "only %d patch-kernel configurations took ks=%d %dx%d" % (ran, ks, H, W)


def gn_ref(x, groups, gamma, beta, eps, silu):
    y = F.group_norm(x.half().float(), groups, gamma, beta, eps)
    if silu:
        y = F.silu(y)
    return y.half().float()  # the normalised operand is staged as fp16


@pytest.mark.parametrize("ks,c1,c2,H,W,silu,eps", [
    (3, 224, 0, 32, 32, True, 1e-5), (3, 448, 224, 16, 16, True, 1e-5), (3, 896, 448, 8, 8, True, 1e-5),
    (3, 896, 896, 4, 4, True, 1e-5), (3, 224, 0, 32, 24, True, 1e-6), (3, 448, 224, 8, 6, True, 1e-5),
    (3, 896, 0, 4, 3, True, 1e-5), (1, 224, 0, 32, 32, False, 1e-6), (1, 448, 0, 16, 12, False, 1e-6),
    (1, 896, 0, 4, 3, False, 1e-6), (3, 128, 0, 64, 64, True, 1e-6)])
def test_pconv_fused_groupnorm_from_group_partials(ctx, ks, c1, c2, H, W, silu, eps):
    """GroupNorm(32) (+ SiLU) of a one- or two-source input folded into the conv: the statistics come from
    upk_groupnorm_stats_nhwc_f16's per-(chunk, group) partials (gni_mode 1); groups may straddle the concat seam."""
    B, cout = 2, 64
    Cc = c1 + c2
    x = rnd(B, Cc, H, W) * 2.0 + 0.5
    gamma, beta = rnd(Cc, seed=3) * 0.5 + 1.0, rnd(Cc, seed=4) * 0.3
    w = rnd(cout, Cc, ks, ks, scale=1 / math.sqrt(ks * ks * Cc))
    b = rnd(cout, scale=0.1)
    ref = F.conv2d(gn_ref(x, 32, gamma, beta, eps, silu), w.half().float(), b, padding=ks // 2)
    x1 = nhwc16(x[:, :c1])
    x2 = nhwc16(x[:, c1:]) if c2 else None
    ws = torch.zeros(ctx.groupnorm_ws_bytes(B, H * W) // 4 + 64, device=DEV)
    ctx._chk(ctx.lib.upk_groupnorm_stats_nhwc_f16(ctx.h, x1.data_ptr(), c1, c1, x2.data_ptr() if c2 else None, c2, c2, B,
                                                  H * W, 32, ws.data_ptr(), ctx._s()))
    ran = 0
    for cfg in range(ctx.lib.upk_pconv_num_configs()):
        y = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
        d = make_desc(ctx, x1, w, b, y, x2=x2)
        d.pc_enable, d.pc_cfg = 1, cfg + 1
        d.gni_mode, d.gni_silu, d.gni_groups, d.gni_eps = 1, int(silu), 32, eps
        if not supported(ctx, d):
            continue
        d.gni_gamma, d.gni_beta = gamma.data_ptr(), beta.data_ptr()
        d.gni_stats1, d.gni_nblk1 = ws.data_ptr(), ctx.lib.upk_groupnorm_chunks(H * W)
        ctx.conv(d)
        torch.cuda.synchronize()
        check(y.permute(0, 3, 1, 2), ref)
        ran += 1
    assert ran >= 1
    # outside the patch kernel the fused form is refused, not silently ignored
    y = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
    d = make_desc(ctx, x1, w, b, y, x2=x2)
    d.gni_mode, d.gni_groups = 1, 32
    with pytest.raises(RuntimeError):
        ctx.conv(d)


@pytest.mark.parametrize("H,W,c,cmid", [(32, 32, 224, 224), (16, 16, 448, 224), (8, 8, 96, 896), (4, 4, 896, 896),
                                        (32, 24, 64, 224), (4, 3, 128, 448)])
def test_pconv_chain_with_channel_partials_and_appended_skip(ctx, H, W, c, cmid):
    """ResBlock shape (openaimodel.py:255-275): conv1 leaves the per-(row block, channel) partials of its output in its
    epilogue, conv2 = conv3x3(SiLU(GroupNorm(h))) + conv1x1(x) as one launch normalises h on the fly from them
    (gni_mode 2) and walks the appended K segment; bitwise reproducible run to run."""
    B, cout = 2, cmid
    x = rnd(B, c, H, W)
    w1 = rnd(cmid, c, 3, 3, scale=1 / math.sqrt(9 * c), seed=1)
    b1 = rnd(cmid, scale=0.1, seed=2)
    gamma, beta = rnd(cmid, seed=3) * 0.5 + 1.0, rnd(cmid, seed=4) * 0.3
    w2 = rnd(cout, cmid, 3, 3, scale=1 / math.sqrt(9 * cmid), seed=5)
    wsk = rnd(cout, c, 1, 1, scale=1 / math.sqrt(c), seed=6)
    b2 = rnd(cout, scale=0.1, seed=7)
    h_ref = F.conv2d(x.half().float(), w1.half().float(), b1, padding=1).half().float()
    ref = F.conv2d(gn_ref(h_ref, 32, gamma, beta, 1e-5, True), w2.half().float(), None, padding=1) \
        + F.conv2d(x.half().float(), wsk.half().float(), None) + b2.view(1, -1, 1, 1)
    xn = nhwc16(x)
    h = torch.zeros(B, H, W, cmid, device=DEV, dtype=torch.float16)
    d1 = make_desc(ctx, xn, w1, b1, h)
    d1.pc_enable = 1
    sws = torch.zeros(ctx.gn_stats_floats(B, d1.n_pad), device=DEV)
    d1.gn_stats_ws, d1.gn_groups = sws.data_ptr(), 32
    assert supported(ctx, d1)
    mode, nblk = ctx.conv_gn_fused(d1)
    assert mode == 2 and nblk >= 1
    ctx.conv(d1)
    wp2, n_pad = ctx.pack_weight(w2.contiguous())
    wpk, _ = ctx.pack_weight(wsk.contiguous())
    wcat = torch.cat([wp2.reshape(-1), wpk.reshape(-1)])
    outs = []
    for rep in range(2):
        y = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
        d2 = make_desc(ctx, h, w2, b2, y)
        d2.w_packed = wcat.data_ptr()
        d2.x3, d2.c3, d2.ld3 = xn.data_ptr(), c, c
        d2.pc_enable = 1
        d2.gni_mode, d2.gni_silu, d2.gni_groups, d2.gni_eps = 2, 1, 32, 1e-5
        assert supported(ctx, d2)
        d2.gni_gamma, d2.gni_beta = gamma.data_ptr(), beta.data_ptr()
        d2.gni_stats1, d2.gni_nblk1, d2.gni_ld1 = sws.data_ptr(), nblk, d1.n_pad
        ctx.conv(d2)
        torch.cuda.synchronize()
        outs.append(y)
    check(h.permute(0, 3, 1, 2), h_ref)
    check(outs[0].permute(0, 3, 1, 2), ref)
    assert torch.equal(outs[0], outs[1])


def test_pconv_output_partials_feed_groupnorm_apply(ctx):
    """The patch kernel's epilogue partials drive upk_groupnorm_apply_nhwc_f16 (mode 2) like an igemm launch's."""
    B, c, cout, H, W = 2, 64, 224, 16, 12
    x = rnd(B, c, H, W)
    w = rnd(cout, c, 3, 3, scale=1 / math.sqrt(9 * c))
    b = rnd(cout, scale=0.1)
    gamma, beta = rnd(cout, seed=3) * 0.5 + 1.0, rnd(cout, seed=4) * 0.3
    y = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
    d = make_desc(ctx, nhwc16(x), w, b, y)
    d.pc_enable = 1
    sws = torch.zeros(ctx.gn_stats_floats(B, d.n_pad), device=DEV)
    d.gn_stats_ws, d.gn_groups = sws.data_ptr(), 32
    mode, nblk = ctx.conv_gn_fused(d)
    assert mode == 2
    ctx.conv(d)
    yn = torch.zeros_like(y)
    ctx._chk(ctx.lib.upk_groupnorm_apply_nhwc_f16(ctx.h, y.data_ptr(), cout, cout, None, 0, 0, B, H * W, 32,
                                                  gamma.data_ptr(), beta.data_ptr(), 1e-5, 1, yn.data_ptr(), cout,
                                                  sws.data_ptr(), 2, nblk, d.n_pad, None, 0, 0, ctx._s()))
    torch.cuda.synchronize()
    ref = F.silu(F.group_norm(y.float().permute(0, 3, 1, 2), 32, gamma, beta, 1e-5))
    check(yn.permute(0, 3, 1, 2), ref)


def test_pconv_geglu_and_nchw_epilogues(ctx):
    """The igemm epilogues behind the patch kernel: fp32 NCHW output of a 3x3 conv with N = 4 (UNet `out`)."""
    B, c, H, W = 2, 224, 32, 32
    x = rnd(B, c, H, W)
    w = rnd(4, c, 3, 3, scale=1 / math.sqrt(9 * c))
    b = rnd(4, scale=0.1)
    ref = F.conv2d(x.half().float(), w.half().float(), b, padding=1)
    y = torch.zeros(B, 4, H, W, device=DEV)
    d = make_desc(ctx, nhwc16(x), w, b, y.view(B, 4, H * W), flags=L.F_OUT_NCHW_F32)
    d.ldy = 0
    d.pc_enable = 1
    assert supported(ctx, d)
    ctx.conv(d)
    torch.cuda.synchronize()
    check(y, ref)

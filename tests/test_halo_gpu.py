"""Halo-patch 3x3 conv family (upgpt_amd/csrc/halo.hip; configurations "hc<NI>p<PF>" of upk_conv_config_name) through
the C ABI: every configuration x split-K against F.conv2d on the feature-map sizes of the UNet levels (64 output pixels
= rows of one image or whole images), two-source concat, the appended 1x1 segment, the plain epilogue's operands,
GroupNorm channel partials, agreement with the wave-specialised family to fp16 rounding, bitwise reproducibility, and the
refusals (stride 2, 1x1, feature maps that are not powers of two, non-plain epilogues unsplit)."""
import math

import pytest
import torch
import torch.nn.functional as F

from upgpt_amd import _lib as L
from test_ops_gpu import DEV, check, conv_ref, make_desc, nhwc16, rnd

pytestmark = pytest.mark.gpu


def hc_cfgs(ctx):
    n = ctx.lib.upk_conv_num_configs()
    out = [(i, ctx.lib.upk_conv_config_name(i).decode()) for i in range(n)]
    return [(i, s) for i, s in out if s.startswith("hc")]


def test_family_is_listed_last(ctx):
    names = [ctx.lib.upk_conv_config_name(i).decode() for i in range(ctx.lib.upk_conv_num_configs())]
    hc = [i for i, s in enumerate(names) if s.startswith("hc")]
    assert len(hc) >= 2 and hc == list(range(hc[0], len(names)))
    assert names[hc[0] - 1].startswith("bt")


def close_to(y, ref16):
    """Against another kernel family's fp16 result: the same fp32 sums in another order — a few fp16 ulps apart."""
    d = (y.float() - ref16.float()).abs()
    tol = 4e-3 * ref16.float().abs().clamp(min=1.0)
    assert bool((d <= tol).all()), "max diff %g" % d.max().item()


@pytest.mark.parametrize("B,cin,cout,hw", [
    (2, 224, 224, (32, 32)),   # level 0: two image rows per tile, 4 x 34 patch, one round of 7 chunks
    (3, 448, 448, (16, 16)),   # level 1: four rows per tile, 6 x 18 patch
    (2, 896, 128, (8, 8)),     # level 2: one image per tile, fragments span two rows; 28 chunks: two slots
    (8, 256, 96, (4, 4)),      # level 3: four images per tile
    (1, 64, 64, (64, 64)),     # one image row per tile (upscale UNet), 3 x 66 patch = 13 groups
    (1, 1344, 64, (16, 16)),   # 42 chunks: six rounds through two slots
    (2, 32, 48, (16, 32)),     # one chunk: fewer items than waves; H != W
])
def test_every_configuration_matches_conv2d(ctx, B, cin, cout, hw):
    H, W = hw
    x = rnd(B, cin, H, W)
    w = rnd(cout, cin, 3, 3, scale=1 / math.sqrt(9 * cin))
    b = rnd(cout, scale=0.1)
    r = rnd(B, H, W, cout, seed=5).half()
    ref = conv_ref(x, w, b) + r.float().permute(0, 3, 1, 2)
    xn = nhwc16(x)
    ws = [i for i in range(ctx.lib.upk_conv_num_configs()) if ctx.lib.upk_conv_config_name(i).decode() == "2x2x2x2k4w3"][0]
    ran = 0
    try:
        yw = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
        ctx.conv_override(ws, 1)
        ctx.conv(make_desc(ctx, xn, w, b, yw, residual=r))
        torch.cuda.synchronize()
        check(yw.permute(0, 3, 1, 2), ref)
        for cfg, name in hc_cfgs(ctx):
            for sk in (1, 2, 3, 7):
                y = torch.full((B, H, W, cout), float("nan"), device=DEV, dtype=torch.float16)
                ctx.conv_override(cfg, sk)
                try:
                    ctx.conv(make_desc(ctx, xn, w, b, y, residual=r))
                except L.UpkError:
                    # (a split with no chunks of its own; 128-pixel tiles need 128 | M and a patch of <= 256 pixels)
                    assert (sk > 1 and (cin // 32 < 2 * sk or sk == 7)) or name.startswith("hc8"), (name, sk)
                    continue
                torch.cuda.synchronize()
                assert torch.isfinite(y).all(), (name, sk)
                check(y.permute(0, 3, 1, 2), ref)
                close_to(y, yw)
                y2 = torch.zeros_like(y)
                ctx.conv(make_desc(ctx, xn, w, b, y2, residual=r))
                torch.cuda.synchronize()
                assert torch.equal(y, y2), (name, sk)  # (fixed summation order)
                ran += 1
    finally:
        ctx.conv_override(-1, 0)
    assert ran >= len(hc_cfgs(ctx))


@pytest.mark.parametrize("c1,c2,c3,c4,cout,hw", [(224, 448, 0, 0, 224, (32, 32)), (448, 0, 448, 224, 448, (16, 16)),
                                                 (96, 64, 32, 64, 112, (8, 8)), (224, 0, 224, 0, 224, (32, 32))])
def test_concat_and_appended_segment(ctx, c1, c2, c3, c4, cout, hw):
    """Decoder ResBlocks: conv3x3 over the concat [h | skip] (x1 | x2) and the second conv with the 1x1 skip projection
    as an appended K segment over (x3 | x4) at the output pixel, timestep row vector and residual in the epilogue."""
    B = 2
    H, W = hw
    xa = rnd(B, c1, H, W, seed=1)
    xb = rnd(B, c2, H, W, seed=2) if c2 else None
    x3 = rnd(B, c3, H, W, seed=3) if c3 else None
    x4 = rnd(B, c4, H, W, seed=4) if c4 else None
    w1 = rnd(cout, c1 + c2, 3, 3, scale=1 / math.sqrt(9 * (c1 + c2)), seed=5)
    b = rnd(cout, scale=0.1, seed=6)
    xin = torch.cat([xa, xb], 1) if c2 else xa
    ref = conv_ref(xin, w1, b)
    wp, n_pad = ctx.pack_weight(w1.contiguous())
    if c3:
        w2 = rnd(cout, c3 + c4, 1, 1, scale=1 / math.sqrt(c3 + c4), seed=7)
        xs = torch.cat([x3, x4], 1) if c4 else x3
        ref = ref + F.conv2d(xs.half().float(), w2.half().float(), None)
        wp2, n_pad2 = ctx.pack_weight(w2.contiguous())
        assert n_pad == n_pad2
        wp = torch.cat([wp.reshape(-1), wp2.reshape(-1)])
    rv = rnd(3, B, cout, seed=8)
    step = torch.tensor([1], dtype=torch.int32, device=DEV)
    ref = ref + rv[1][:, :, None, None]
    res = rnd(B, H, W, cout, seed=9).half()
    ref = ref + res.float().permute(0, 3, 1, 2)
    keep = [nhwc16(t) if t is not None else None for t in (xa, xb, x3, x4)]
    ran = 0
    try:
        for cfg, name in hc_cfgs(ctx):
            for sk in (1, 2, 3):
                y = torch.full((B, H, W, cout), float("nan"), device=DEV, dtype=torch.float16)
                d = make_desc(ctx, keep[0], w1, b, y, x2=keep[1], residual=res, rowvec=rv, rv_bs=cout, rv_ss=B * cout, step=step)
                d.w_packed = wp.data_ptr()
                if c3:
                    d.x3, d.c3, d.ld3 = keep[2].data_ptr(), c3, c3
                if c4:
                    d.x4, d.c4, d.ld4 = keep[3].data_ptr(), c4, c4
                ctx.conv_override(cfg, sk)
                try:
                    ctx.conv(d)
                except L.UpkError:
                    assert sk > 1 or name.startswith("hc8"), (name, sk)
                    continue
                torch.cuda.synchronize()
                assert torch.isfinite(y).all(), (name, sk)
                check(y.permute(0, 3, 1, 2), ref, tol=6e-3)
                ran += 1
    finally:
        ctx.conv_override(-1, 0)
    assert ran >= len(hc_cfgs(ctx))


def test_groupnorm_partials_from_the_epilogue(ctx):
    """upk_conv_desc.gn_stats_ws on an unsplit launch: apply-only GroupNorm on the per-(64-pixel tile, channel) partials
    equals the two-pass GroupNorm of the stored tensor (bit for bit: same fold order in the apply pass)."""
    B, H, W, cin, cout = 2, 32, 32, 64, 224
    x = rnd(B * H * W, cin).half()
    w = rnd(cout, cin, 3, 3, scale=1 / math.sqrt(9 * cin))
    b = rnd(cout, scale=0.1)
    wp, n_pad = ctx.pack_weight(w)
    bp = torch.zeros(n_pad, device=DEV); bp[:cout] = b
    y = torch.zeros(B * H * W, cout, device=DEV, dtype=torch.float16)
    sws = torch.full((ctx.gn_stats_floats(B, n_pad),), float("nan"), device=DEV)
    d = L.ConvDesc()
    d.x1 = x.data_ptr(); d.c1 = cin; d.ld1 = cin; d.batch = B; d.in_h = H; d.in_w = W; d.ksize = 3; d.stride = 1
    d.w_packed = wp.data_ptr(); d.n_out = cout; d.n_pad = n_pad; d.bias = bp.data_ptr(); d.y = y.data_ptr(); d.ldy = cout
    d.gn_stats_ws = sws.data_ptr(); d.gn_groups = 32
    gamma, beta = 1 + 0.1 * rnd(cout, seed=2), 0.1 * rnd(cout, seed=3)
    ws2 = torch.zeros(ctx.groupnorm_ws_bytes(B, H * W) // 4, device=DEV)
    ran = 0
    try:
        for cfg, name in hc_cfgs(ctx):
            ctx.conv_override(cfg, 1)
            mode, nblk = ctx.conv_gn_fused(d)
            assert mode == 2 and nblk == H * W // (16 * int(name[2])), name
            y.zero_(); sws.fill_(float("nan"))
            ctx.conv(d)
            app, full = torch.zeros_like(y), torch.zeros_like(y)
            ctx._chk(ctx.lib.upk_groupnorm_apply_nhwc_f16(
                ctx.h, y.data_ptr(), cout, cout, None, 0, 0, B, H * W, 32, gamma.data_ptr(), beta.data_ptr(), 1e-5, 1,
                app.data_ptr(), cout, sws.data_ptr(), 2, nblk, n_pad, None, 0, 0, ctx._s()))
            ctx._chk(ctx.lib.upk_groupnorm_nhwc_f16(
                ctx.h, y.data_ptr(), cout, cout, None, 0, 0, B, H * W, 32, gamma.data_ptr(), beta.data_ptr(), 1e-5, 1,
                full.data_ptr(), cout, ws2.data_ptr(), ctx._s()))
            torch.cuda.synchronize()
            ref = F.silu(F.group_norm(y.float().view(B, H * W, cout).permute(0, 2, 1), 32, gamma, beta, 1e-5)).permute(0, 2, 1)
            check(app.view(B, H * W, cout), ref, tol=5e-3)
            d1 = (app.float() - full.float()).abs().max().item()
            assert d1 <= 2e-3 * max(1.0, full.float().abs().max().item()), (name, d1)
            ran += 1
    finally:
        ctx.conv_override(-1, 0)
    assert ran == len(hc_cfgs(ctx))


def test_refusals(ctx):
    """Outside the family's domain the library says so (UPK_ESHAPE), it never runs something else silently."""
    cfg = hc_cfgs(ctx)[0][0]
    B, cin, cout = 2, 64, 64

    def desc(H, W, ks=3, stride=1, flags=0):
        x = rnd(B, cin, H, W)
        w = rnd(cout, cin, ks, ks, scale=0.05)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = torch.zeros(B, Ho, Wo, cout, device=DEV, dtype=torch.float16)
        return make_desc(ctx, nhwc16(x), w, None, y, stride=stride, flags=flags)

    try:
        ctx.conv_override(cfg, 1)
        ctx.conv(desc(16, 16))  # (in the domain)
        for bad in (desc(16, 24), desc(12, 16), desc(16, 16, ks=1), desc(16, 16, stride=2), desc(16, 16, flags=L.F_SILU),
                    desc(8, 8, flags=L.F_UPSAMPLE2X)):
            with pytest.raises(L.UpkError):
                ctx.conv(bad)
        ctx.conv_override(cfg, 2)
        ctx.conv(desc(16, 16, flags=L.F_SILU))  # (split: the reduce pass runs the general epilogue)
        ctx.conv_override(cfg, 3)
        with pytest.raises(L.UpkError):
            ctx.conv(desc(16, 16))  # (two chunks cannot be split three ways)
    finally:
        ctx.conv_override(-1, 0)
    torch.cuda.synchronize()


def channel_partials(x_rows, B, hw, nblk, ld):
    """upk_conv_desc.gn_stats_ws mode 2 layout [B][nblk][2][ld] of an NHWC fp16 tensor [B * hw, C]: per-(row block,
    channel) sum and sum of squares (what a producer conv's epilogue leaves)."""
    C_ = x_rows.shape[1]
    xf = x_rows.float().view(B, nblk, hw // nblk, C_)
    st = torch.zeros(B, nblk, 2, ld, device=DEV)
    st[:, :, 0, :C_] = xf.sum(2)
    st[:, :, 1, :C_] = (xf * xf).sum(2)
    return st.contiguous()


@pytest.mark.parametrize("c1,c2,c3,cout,hw,nb1,nb2,silu", [
    (224, 0, 0, 224, (32, 32), 16, 0, 1),    # ResBlock in_layers / out_layers at level 0
    (224, 224, 0, 224, (32, 32), 16, 32, 1),  # decoder ResBlock: GroupNorm over the concat [h | skip], two producers
    (448, 0, 224, 448, (16, 16), 4, 0, 1),   # level 1 out_layers with the appended 1x1 skip projection (not normalised)
    (96, 64, 0, 80, (16, 16), 2, 4, 0),      # groups of 5 channels straddling the concat seam; no SiLU
    (64, 0, 0, 64, (64, 64), 32, 0, 1),      # one image row per tile
])
def test_input_groupnorm_in_the_patch_fill(ctx, c1, c2, c3, cout, hw, nb1, nb2, silu):
    """upk_conv_desc.gni_*: GroupNorm(+SiLU) -> conv3x3 as ONE launch (openaimodel.py:203-206, 227-233) is bit-identical
    to upk_groupnorm_apply_nhwc_f16 followed by the same halo-patch configuration on the normalised tensor, and matches
    F.group_norm -> F.silu -> F.conv2d."""
    B = 3
    H, W = hw
    C_ = c1 + c2
    xa = (rnd(B * H * W, c1, seed=1) * 1.5 + 0.3).half()
    xb = (rnd(B * H * W, c2, seed=2) * 0.7 - 0.2).half() if c2 else None
    x3 = rnd(B * H * W, c3, seed=3).half() if c3 else None
    ld1, ld2 = c1 + 32, c2 + 64
    st1 = channel_partials(xa, B, H * W, nb1, ld1)
    st2 = channel_partials(xb, B, H * W, nb2, ld2) if c2 else None
    gamma, beta = 1 + 0.2 * rnd(C_, seed=4), 0.2 * rnd(C_, seed=5)
    w1 = rnd(cout, C_, 3, 3, scale=1 / math.sqrt(9 * C_), seed=6)
    b = rnd(cout, scale=0.1, seed=7)
    wp, n_pad = ctx.pack_weight(w1.contiguous())
    res = rnd(B, H, W, cout, seed=9).half()
    xcat = torch.cat([xa, xb], 1) if c2 else xa
    xt = xcat.float().view(B, H * W, C_).permute(0, 2, 1).reshape(B, C_, H, W)
    gn = F.group_norm(xt, 32 if C_ % 32 == 0 else 16, gamma, beta, 1e-5)
    groups = 32 if C_ % 32 == 0 else 16
    ref = conv_ref(F.silu(gn) if silu else gn, w1, b) + res.float().permute(0, 3, 1, 2)
    if c3:
        w2 = rnd(cout, c3, 1, 1, scale=1 / math.sqrt(c3), seed=8)
        ref = ref + F.conv2d(x3.float().view(B, H, W, c3).permute(0, 3, 1, 2), w2.half().float(), None)
        wp2, _ = ctx.pack_weight(w2.contiguous())
        wp = torch.cat([wp.reshape(-1), wp2.reshape(-1)])
    # the two-launch form: apply pass on the partials, then the conv on the normalised tensor
    xn = torch.zeros(B * H * W, C_, device=DEV, dtype=torch.float16)
    ctx._chk(ctx.lib.upk_groupnorm_apply_nhwc_f16(
        ctx.h, xa.data_ptr(), c1, c1, xb.data_ptr() if c2 else None, c2, c2, B, H * W, groups, gamma.data_ptr(),
        beta.data_ptr(), 1e-5, silu, xn.data_ptr(), C_, st1.data_ptr(), 2, nb1, ld1, st2.data_ptr() if c2 else None, nb2, ld2,
        ctx._s()))

    def base_desc(y, x1t, x2t):
        d = make_desc(ctx, x1t.view(B, H, W, -1), w1, b, y, x2=x2t.view(B, H, W, -1) if x2t is not None else None, residual=res)
        d.w_packed = wp.data_ptr()
        if c3:
            d.x3, d.c3, d.ld3 = x3.data_ptr(), c3, c3
        return d

    ran = 0
    try:
        for cfg, name in hc_cfgs(ctx):
            ctx.conv_override(cfg, 1)
            y0 = torch.full((B, H, W, cout), float("nan"), device=DEV, dtype=torch.float16)
            d0 = base_desc(y0, xa, xb)  # (same sources split as the fused launch: same K order, bit-comparable)
            d0.x1, d0.ld1 = xn.data_ptr(), C_
            if c2:
                d0.x2, d0.ld2 = xn.data_ptr() + 2 * c1, C_
            try:
                ctx.conv(d0)
            except L.UpkError:
                assert name.startswith("hc8"), name
                continue
            y1 = torch.full((B, H, W, cout), float("nan"), device=DEV, dtype=torch.float16)
            d = base_desc(y1, xa, xb)
            d.gni_stats1, d.gni_nblk1, d.gni_ld1 = st1.data_ptr(), nb1, ld1
            if c2:
                d.gni_stats2, d.gni_nblk2, d.gni_ld2 = st2.data_ptr(), nb2, ld2
            d.gni_gamma, d.gni_beta, d.gni_eps = gamma.data_ptr(), beta.data_ptr(), 1e-5
            d.gni_groups, d.gni_silu = groups, silu
            if not ctx.conv_gn_input(d):
                with pytest.raises(L.UpkError):
                    ctx.conv(d)
                continue
            ctx.conv(d)
            torch.cuda.synchronize()
            assert torch.isfinite(y1).all(), name
            assert torch.equal(y0, y1), (name, (y0.float() - y1.float()).abs().max().item())
            check(y1.permute(0, 3, 1, 2), ref, tol=8e-3)
            ran += 1
        # any other family: the library says no, and refuses the launch instead of convolving the raw tensor
        ctx.conv_override(0, 1)
        assert not ctx.conv_gn_input(d)
        with pytest.raises(L.UpkError):
            ctx.conv(d)
    finally:
        ctx.conv_override(-1, 0)
    assert ran >= 1


def test_input_groupnorm_refused_when_the_patch_needs_two_slots(ctx):
    """A K range that does not fit one LDS slot (28 chunks at 8x8) keeps the two-launch form: upk_conv_gn_input says 0."""
    B, H, W, cin, cout = 2, 8, 8, 896, 64
    x = rnd(B * H * W, cin).half()
    st = channel_partials(x, B, H * W, 1, cin)
    w = rnd(cout, cin, 3, 3, scale=0.02)
    y = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
    gamma, beta = rnd(cin, seed=1), rnd(cin, seed=2)
    d = make_desc(ctx, x.view(B, H, W, cin), w, None, y)
    d.gni_stats1, d.gni_nblk1, d.gni_ld1 = st.data_ptr(), 1, cin
    d.gni_gamma, d.gni_beta, d.gni_eps, d.gni_groups, d.gni_silu = gamma.data_ptr(), beta.data_ptr(), 1e-5, 32, 1
    try:
        ctx.conv_override(hc_cfgs(ctx)[0][0], 1)
        assert not ctx.conv_gn_input(d)
    finally:
        ctx.conv_override(-1, 0)

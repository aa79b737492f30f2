"""Big-tile conv family (upgpt_amd/csrc/bigtile.hip; configurations "bt<MI>x<NI>x<WM>x<WN>n<NBUF>" of
upk_conv_config_name) through the C ABI: every configuration against F.conv2d on VAE-decoder-like shapes (many tiles
per CU, ragged tile edges, two-source concat, stride 2, nearest-2x upsample as four phase convs, split-K), bit-identity
with the wave-specialised family for the same split, the GroupNorm channel partials / LayerNorm row sums of the plain epilogue, the
general epilogue of the 4-wave configurations and the refusals (8-wave + non-plain unsplit, appended K segment)."""
import math

import pytest
import torch
import torch.nn.functional as F

from upgpt_amd import _lib as L
from test_ops_gpu import DEV, check, conv_ref, make_desc, nhwc16, rnd, _phase_weights

pytestmark = pytest.mark.gpu


def bt_cfgs(ctx):
    n = ctx.lib.upk_conv_num_configs()
    out = [(i, ctx.lib.upk_conv_config_name(i).decode()) for i in range(n)]
    return [(i, s) for i, s in out if s.startswith("bt")]


def test_family_is_listed_behind_the_others(ctx):
    names = [ctx.lib.upk_conv_config_name(i).decode() for i in range(ctx.lib.upk_conv_num_configs())]
    bt = [i for i, s in enumerate(names) if s.startswith("bt")]
    assert len(bt) >= 4 and bt == list(range(bt[0], len(names)))  # (families are appended, never inserted)
    assert all(s.startswith("as") for s in names[bt[0] - 4:bt[0]])


@pytest.mark.parametrize("B,cin,cout,hw,ks", [
    (2, 128, 128, (48, 40), 3),   # N = one 128-wide tile, M = 3840 = 15 tiles of 256: ragged last tile
    (1, 256, 256, (64, 36), 3),   # 256 x 256 tiles, 9 of them
    (2, 512, 512, (16, 24), 3),   # K = 4608: 144 stages
    (3, 64, 320, (20, 20), 1),    # 1x1, N not a tile multiple
    (1, 96, 132, (9, 7), 3),      # smaller than one tile in both dimensions
    (2, 224, 224, (32, 20), 3),   # the 7 * 32-channel family: N = one 224-wide tile (14 row groups dealt unevenly to 4 / 8 waves)
    (1, 448, 448, (16, 16), 1),   # ... N = two 224-wide / four 112-wide tiles, 1x1
])
def test_every_configuration_matches_conv2d(ctx, B, cin, cout, hw, ks):
    H, W = hw
    x = rnd(B, cin, H, W)
    w = rnd(cout, cin, ks, ks, scale=1 / math.sqrt(ks * ks * cin))
    b = rnd(cout, scale=0.1)
    r = rnd(B, H, W, cout, seed=5).half()
    ref = conv_ref(x, w, b) + r.float().permute(0, 3, 1, 2)
    xn = nhwc16(x)
    base = None
    try:
        for cfg, name in bt_cfgs(ctx):
            for sk in (1, 2, 3):
                y = torch.full((B, H, W, cout), float("nan"), device=DEV, dtype=torch.float16)
                ctx.conv_override(cfg, sk)
                try:
                    ctx.conv(make_desc(ctx, xn, w, b, y, residual=r))
                except L.UpkError:
                    assert sk > 1, name  # (only a split finer than the K loop allows may be refused)
                    continue
                torch.cuda.synchronize()
                assert torch.isfinite(y).all(), (name, sk)
                check(y.permute(0, 3, 1, 2), ref)
                if sk == 1:  # same fp32 summation order in every unsplit configuration of the family
                    base = y if base is None else base
                    assert torch.equal(y, base), name
        # ... and the same bits as the wave-specialised family's unsplit launch (same K order, same epilogue arithmetic)
        ws = [i for i in range(ctx.lib.upk_conv_num_configs()) if ctx.lib.upk_conv_config_name(i).decode() == "4x4x2x2k2w3"][0]
        y = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
        ctx.conv_override(ws, 1)
        ctx.conv(make_desc(ctx, xn, w, b, y, residual=r))
        torch.cuda.synchronize()
        assert torch.equal(y, base)
    finally:
        ctx.conv_override(-1, 0)


def test_concat_stride2_upsample_and_phases(ctx):
    """The loader paths the UNet / VAE use besides the plain 3x3: two-source channel concat, stride 2, nearest-2x
    upsample in the loader, and the same upsample as four 2x2 phase convs (grid.y = phase)."""
    B, H, W = 2, 24, 20
    xa, xb = rnd(B, 64, H, W, seed=1), rnd(B, 96, H, W, seed=2)
    w = rnd(160, 160, 3, 3, scale=1 / math.sqrt(9 * 160), seed=3)
    b = rnd(160, scale=0.1, seed=4)
    try:
        for cfg, name in bt_cfgs(ctx):
            ctx.conv_override(cfg, 1)
            y = torch.zeros(B, H, W, 160, device=DEV, dtype=torch.float16)
            ctx.conv(make_desc(ctx, nhwc16(xa), w, b, y, x2=nhwc16(xb)))
            torch.cuda.synchronize()
            check(y.permute(0, 3, 1, 2), conv_ref(torch.cat([xa, xb], 1), w, b))
            x = torch.cat([xa, xb], 1)
            y = torch.zeros(B, H // 2, W // 2, 160, device=DEV, dtype=torch.float16)
            ctx.conv(make_desc(ctx, nhwc16(x), w, b, y, stride=2))
            torch.cuda.synchronize()
            check(y.permute(0, 3, 1, 2), conv_ref(x, w, b, stride=2))
            ref = conv_ref(x, w, b, ups=True)
            y = torch.zeros(B, 2 * H, 2 * W, 160, device=DEV, dtype=torch.float16)
            ctx.conv(make_desc(ctx, nhwc16(x), w, b, y, flags=L.F_UPSAMPLE2X))
            torch.cuda.synchronize()
            check(y.permute(0, 3, 1, 2), ref)
            y2 = torch.zeros_like(y)
            d = make_desc(ctx, nhwc16(x), w, b, y2, flags=L.F_UPSAMPLE2X)
            wph, n_pad = _phase_weights(ctx, w)
            assert d.n_pad == n_pad
            d.w_phase = wph.data_ptr()
            ctx.conv(d)
            torch.cuda.synchronize()
            check(y2.permute(0, 3, 1, 2), ref)
    finally:
        ctx.conv_override(-1, 0)


def test_groupnorm_partials_from_the_epilogue(ctx):
    """upk_conv_desc.gn_stats_ws on an unsplit launch whose M tiles lie inside one sample: apply-only GroupNorm on the
    per-(M tile, channel) partials equals the two-pass GroupNorm of the stored tensor."""
    B, H, W, cin, cout = 2, 32, 32, 64, 256
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
        for cfg, name in bt_cfgs(ctx):
            ctx.conv_override(cfg, 1)
            mode, nblk = ctx.conv_gn_fused(d)
            assert mode in (0, 2), name
            if mode != 2:
                continue
            y.zero_(); sws.fill_(float("nan"))
            ctx.conv(d)
            app, full = torch.zeros_like(y), torch.zeros_like(y)
            ctx._chk(ctx.lib.upk_groupnorm_apply_nhwc_f16(
                ctx.h, y.data_ptr(), cout, cout, None, 0, 0, B, H * W, 32, gamma.data_ptr(), beta.data_ptr(), 1e-5, 1,
                app.data_ptr(), cout, sws.data_ptr(), 2, nblk, n_pad, None, 0, 0, ctx._s()))
            ctx.groupnorm(y, cout, cout, None, 0, 0, B, H * W, 32, gamma, beta, 1e-5, True, full, cout, ws2)
            torch.cuda.synchronize()
            assert (app.float() - full.float()).abs().max().item() < 4e-3, name
            ran += 1
    finally:
        ctx.conv_override(-1, 0)
    assert ran >= 3


@pytest.mark.parametrize("flags,f32", [(L.F_SILU, False), (L.F_QUICKGELU, False), (0, True)])
def test_other_epilogues_and_the_refused_appended_segment(ctx, flags, f32):
    """The general epilogue (SiLU, quick-GELU, fp32 output ...) exists for the 4-wave configurations (their own
    instantiations) and for every configuration when K is split (the reduce pass runs it); the 8-wave configurations
    refuse it unsplit.  The appended 1x1 K segment lives in the wave-specialised loader only: always refused."""
    B, H, W, cin, cout = 1, 16, 20, 64, 160
    x = rnd(B, cin, H, W)
    w = rnd(cout, cin, 3, 3, scale=1 / math.sqrt(9 * cin))
    b = rnd(cout, scale=0.1)
    ref = conv_ref(x, w, b)
    if flags & L.F_SILU:
        ref = F.silu(ref)
    if flags & L.F_QUICKGELU:
        ref = ref * torch.sigmoid(1.702 * ref)
    ran = refused = 0
    try:
        for cfg, name in bt_cfgs(ctx):
            for sk in (1, 2):
                y = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float32 if f32 else torch.float16)
                ctx.conv_override(cfg, sk)
                try:
                    ctx.conv(make_desc(ctx, nhwc16(x), w, b, y, flags=flags | (L.F_OUT_F32 if f32 else 0)))
                except L.UpkError:
                    assert sk == 1, name
                    refused += 1
                    continue
                torch.cuda.synchronize()
                check(y.permute(0, 3, 1, 2), ref)
                ran += 1
        assert ran >= len(bt_cfgs(ctx)) + 3 and refused >= 1, (ran, refused)
        x3 = nhwc16(rnd(B, 32, H, W, seed=9))
        y = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
        d = make_desc(ctx, nhwc16(x), w, b, y)
        d.x3, d.c3, d.ld3 = x3.data_ptr(), 32, 32
        for sk in (1, 2):
            ctx.conv_override(bt_cfgs(ctx)[0][0], sk)
            with pytest.raises(L.UpkError):
                ctx.conv(d)
    finally:
        ctx.conv_override(-1, 0)


def test_geglu_and_layernorm_rows(ctx):
    """Epilogues of the transformer Linears: GEGLU (bit-identical to the wave-specialised family's), LayerNorm row sums
    of a plain output for a folded-LayerNorm consumer (<= 8 column slots)."""
    import ctypes as ct
    from test_ops_gpu import geglu_row_map
    M, C, N = 600, 128, 512
    x = rnd(M, C).half()
    names = [ctx.lib.upk_conv_config_name(i).decode() for i in range(ctx.lib.upk_conv_num_configs())]
    try:
        w = rnd(N, C, scale=1 / math.sqrt(C)); b = rnd(N, scale=0.1)
        rm = geglu_row_map(N // 2).to(DEV)
        outs = []
        for cfg in [names.index("2x4x4x1k2w3")] + [c for c, _ in bt_cfgs(ctx)]:
            y = torch.zeros(1, M, 1, N // 2, device=DEV, dtype=torch.float16)
            d = make_desc(ctx, x.view(1, M, 1, C), w.view(N, C, 1, 1), b, y, flags=L.F_GEGLU, n_out=N // 2, row_map=rm)
            ctx.conv_override(cfg, 1)
            try:
                ctx.conv(d)
            except L.UpkError:
                continue  # (8-wave configurations: plain epilogues only)
            torch.cuda.synchronize()
            outs.append(y.clone())
        v, g = (x.float() @ w.half().float().t() + b).chunk(2, dim=1)
        check(outs[0].view(M, N // 2), v * F.gelu(g))
        assert len(outs) >= 4
        for o in outs[1:]:
            assert torch.equal(o, outs[0])
        w2 = rnd(256, C, scale=1 / math.sqrt(C)); b2 = rnd(256, scale=0.1)
        took = 0
        for cfg, name in bt_cfgs(ctx):
            y = torch.zeros(1, M, 1, 256, device=DEV, dtype=torch.float16)
            rows = torch.zeros(8, M, 2, device=DEV)
            d = make_desc(ctx, x.view(1, M, 1, C), w2.view(256, C, 1, 1), b2, y)
            d.ln_rows_out = rows.data_ptr()
            ctx.conv_override(cfg, 1)
            slots = ct.c_int(0)
            ctx._chk(ctx.lib.upk_conv_ln_rows(ctx.h, ct.byref(d), ct.byref(slots)))
            assert 0 <= slots.value <= 8, name
            ctx.conv(d)
            torch.cuda.synchronize()
            yf = y.view(M, 256).float()
            check(yf, x.float() @ w2.half().float().t() + b2)
            if slots.value:
                got = rows[:slots.value].sum(0)
                assert torch.allclose(got[:, 0], yf.sum(1), rtol=1e-3, atol=1e-2), name
                assert torch.allclose(got[:, 1], (yf * yf).sum(1), rtol=1e-3, atol=1e-2), name
                took += 1
        assert took >= 3
    finally:
        ctx.conv_override(-1, 0)


def test_cost_model_path_on_chip_filling_and_small_launches(ctx):
    """Untuned shapes: a VAE-sized plain conv may go to a big-tile configuration (>= one tile per CU), a UNet-sized one
    never does; whatever the cost model picks must be right."""
    for (B, H, W, c) in ((2, 128, 128, 128), (2, 16, 16, 128)):
        x = rnd(B, c, H, W)
        w = rnd(c, c, 3, 3, scale=1 / math.sqrt(9 * c))
        b = rnd(c, scale=0.1)
        y = torch.zeros(B, H, W, c, device=DEV, dtype=torch.float16)
        ctx.conv(make_desc(ctx, nhwc16(x), w, b, y))
        torch.cuda.synchronize()
        check(y.permute(0, 3, 1, 2), conv_ref(x, w, b))

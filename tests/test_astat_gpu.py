"""A-stationary GEMM family (upgpt_amd/csrc/astat.hip; configurations "as<MI>x<NI>p<PF>" of upk_conv_config_name)
through the C ABI against plain PyTorch fp32 references: every configuration x passes-per-workgroup on the UNet's
Linear shapes, every epilogue the family takes over from the implicit-GEMM kernels."""
import math

import pytest
import torch
import torch.nn.functional as F

from upgpt_amd import _lib as L
from test_ops_gpu import DEV, check, geglu_row_map, rnd

pytestmark = pytest.mark.gpu


def as_cfgs(ctx):
    n = ctx.lib.upk_conv_num_configs()
    return [(i, ctx.lib.upk_conv_config_name(i).decode()) for i in range(n)
            if ctx.lib.upk_conv_config_name(i).decode().startswith("as")]


def lin_desc(x, wp, n_pad, n_out, y, bias=None, flags=0):
    d = L.ConvDesc()
    d.x1 = x.data_ptr(); d.c1 = x.shape[-1]; d.ld1 = x.shape[-1]
    d.batch = 1; d.in_h = x.shape[0]; d.in_w = 1; d.ksize = 1; d.stride = 1
    d.w_packed = wp.data_ptr(); d.n_out = n_out; d.n_pad = n_pad
    if bias is not None:
        d.bias = bias.data_ptr()
    d.y = y.data_ptr(); d.ldy = y.shape[-1]; d.flags = flags
    d._keep = (x, wp, y, bias)
    return d


def sweep(ctx, d, y, ref, tol=2e-2, ppws=(0, 1, 2, 3, 4, 6, 8), need=1, fresh=None):
    """Runs `d` on every A-stationary configuration x passes-per-workgroup that accepts it; returns how many ran."""
    ran = 0
    try:
        for cfg, name in as_cfgs(ctx):
            for ppw in ppws:
                ctx.conv_override(cfg, ppw)
                y.zero_()
                if fresh is not None:
                    fresh()
                try:
                    ctx.conv(d)
                except L.UpkError:
                    continue
                torch.cuda.synchronize()
                try:
                    check(y, ref, tol=tol)
                except AssertionError as e:
                    raise AssertionError("%s ppw=%d: %s" % (name, ppw, e)) from None
                ran += 1
    finally:
        ctx.conv_override(-1, 0)
    assert ran >= need, "only %d A-stationary launches ran" % ran
    return ran


@pytest.mark.parametrize("M,K,N", [(8192, 224, 768), (2048, 448, 1536), (512, 896, 3072), (600, 256, 224), (8192, 256, 224),
                                   (100, 512, 1020), (37, 224, 28), (128, 1024, 896), (2048, 1792, 448)])
def test_astat_bias_residual(ctx, M, K, N):
    x = rnd(M, K).half()
    w = rnd(N, K, scale=1 / math.sqrt(K))
    b = rnd(N, scale=0.1)
    res = rnd(M, N, seed=5).half()
    ref = x.float() @ w.half().float().t() + b + res.float()
    wp, n_pad = ctx.pack_weight(w)
    bp = torch.zeros(n_pad, device=DEV); bp[:N] = b
    ld = (N + 3) // 4 * 4
    y = torch.zeros(M, ld, device=DEV, dtype=torch.float16)
    d = lin_desc(x, wp, n_pad, N, y, bp)
    resp = torch.zeros(M, ld, device=DEV, dtype=torch.float16); resp[:, :N] = res
    d.residual = resp.data_ptr(); d.ld_res = ld
    refp = torch.zeros(M, ld, device=DEV); refp[:, :N] = ref
    sweep(ctx, d, y, refp, need=4)


@pytest.mark.parametrize("M,dm", [(8192, 224), (2048, 448), (512, 896), (100, 256)])
def test_astat_geglu_with_folded_layernorm(ctx, M, dm):
    """GEGLU projection behind norm3 (attention.py:42-44, 215): both fragment pairings (NI = 2: value / gate fragments
    32 columns apart; NI = 4: one 64-column block per wave), statistics from the resident tile."""
    g = torch.Generator(device="cpu").manual_seed(3)
    x = (torch.randn(M, dm, generator=g) * (0.5 + 2 * torch.rand(M, 1, generator=g)) + 3 * torch.randn(M, 1, generator=g)).to(DEV).half()
    inner = 4 * dm
    gamma, beta = 1 + 0.2 * rnd(dm, seed=2), 0.1 * rnd(dm, seed=3)
    w = rnd(2 * inner, dm, scale=1 / math.sqrt(dm), seed=4)
    b = rnd(2 * inner, scale=0.1, seed=5)
    h = F.layer_norm(x.float(), (dm,), gamma, beta, 1e-5) @ w.t() + b
    ref = h[:, :inner] * F.gelu(h[:, inner:])
    rm = geglu_row_map(inner).to(DEV)
    wf = (w * gamma[None, :]).contiguous()
    wp, n_pad = ctx.pack_weight(wf, row_map=rm)
    bf, u = (b + w @ beta)[rm.long()].contiguous(), wf.half().float().sum(dim=1)[rm.long()].contiguous()
    y = torch.zeros(M, inner, device=DEV, dtype=torch.float16)
    d = lin_desc(x, wp, n_pad, inner, y, bf, L.F_GEGLU)
    d.ln_colsum = u.data_ptr(); d.ln_eps = 1e-5; d.ln_dim = dm
    d._k2 = (u,)
    ran = sweep(ctx, d, y, ref, tol=6e-3, need=4)
    # plain GEGLU (no LayerNorm) too
    wp2, _ = ctx.pack_weight(w, row_map=rm)
    h2 = x.float() @ w.half().float().t() + b
    d2 = lin_desc(x, wp2, n_pad, inner, y, b[rm.long()].contiguous(), L.F_GEGLU)
    sweep(ctx, d2, y, h2[:, :inner] * F.gelu(h2[:, inner:]), tol=6e-3, need=4)
    assert ran >= 4


def test_astat_two_sources_and_appended_segment(ctx):
    """[x1 | x2] along K plus an appended (x3 | x4) segment: all of them are chunks of the resident tile."""
    M, c1, c2, c3, c4, N = 1000, 448, 224, 192, 32, 448
    xs = [rnd(M, c, seed=i).half() for i, c in enumerate((c1, c2, c3, c4))]
    w = rnd(N, c1 + c2, scale=1 / math.sqrt(c1 + c2), seed=9)
    w2 = rnd(N, c3 + c4, scale=1 / math.sqrt(c3 + c4), seed=10)
    b = rnd(N, scale=0.1)
    ref = torch.cat(xs[:2], 1).float() @ w.half().float().t() + torch.cat(xs[2:], 1).float() @ w2.half().float().t() + b
    wp, n_pad = ctx.pack_weight(w)
    wq, _ = ctx.pack_weight(w2)
    wa = torch.cat([wp.reshape(-1), wq.reshape(-1)])
    y = torch.zeros(M, N, device=DEV, dtype=torch.float16)
    d = lin_desc(xs[0], wa, n_pad, N, y, b)
    d.x2 = xs[1].data_ptr(); d.c2 = c2; d.ld2 = c2
    d.x3 = xs[2].data_ptr(); d.c3 = c3; d.ld3 = c3
    d.x4 = xs[3].data_ptr(); d.c4 = c4; d.ld4 = c4
    d._k2 = xs
    sweep(ctx, d, y, ref, need=4)  # (448 + 224 + 224) / 32 = 28 chunks: the ring-of-7 configurations


def test_astat_qkv_with_transposed_v_and_row_sums(ctx):
    """q|k|v projection with the V^T tail (attention.hip's operand layout) behind a folded LayerNorm whose row
    statistics come from the producer's row sums (upk_conv_desc.ln_rows_in)."""
    B, T, C, heads, dp = 2, 256, 448, 8, 64
    M, hd = B * T, heads * dp
    x = (rnd(M, C, seed=1) * 2 + 1).half()
    w = rnd(3 * hd, C, scale=1 / math.sqrt(C), seed=2)
    gamma, beta = 1 + 0.2 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    ref = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5) @ w.t()
    wf = (w * gamma[None, :]).contiguous()
    wp, n_pad = ctx.pack_weight(wf)
    bf, u = (w @ beta).contiguous(), wf.half().float().sum(dim=1).contiguous()
    y = torch.zeros(M, 2 * hd, device=DEV, dtype=torch.float16)
    vt = torch.zeros(B, heads, dp, T, device=DEV, dtype=torch.float16)
    rows = torch.zeros(8, M, 2, device=DEV)
    rows[0, :, 0] = x.float().sum(1)
    rows[0, :, 1] = (x.float() ** 2).sum(1)
    d = lin_desc(x, wp, n_pad, 2 * hd, y, bf)
    d.vt = vt.data_ptr(); d.vt_from = 2 * hd; d.vt_heads = heads; d.vt_dhead = dp; d.vt_ld = T; d.vt_tokens = T
    d.ln_colsum = u.data_ptr(); d.ln_eps = 1e-5; d.ln_dim = C
    d.ln_rows_in = rows.data_ptr(); d.ln_rows_slots = 1
    d._k2 = (vt, rows, u)
    ran = 0
    try:
        for cfg, name in as_cfgs(ctx):
            for ppw in (0, 1, 3):
                ctx.conv_override(cfg, ppw)
                y.zero_(); vt.zero_()
                try:
                    ctx.conv(d)
                except L.UpkError:
                    continue
                torch.cuda.synchronize()
                check(y, ref[:, : 2 * hd], tol=6e-3)
                v = ref[:, 2 * hd:].reshape(B, T, heads, dp).permute(0, 2, 3, 1)
                check(vt, v, tol=6e-3)
                ran += 1
    finally:
        ctx.conv_override(-1, 0)
    assert ran >= 4


def test_astat_groupnorm_partials_and_layernorm_row_sums(ctx):
    """The by-products other launches consume: per-(row block, channel) GroupNorm partials (gn_stats_ws, mode 2) and the
    LayerNorm row sums of the output (ln_rows_out), bitwise equal between configurations that tile M alike."""
    B, HW, K, N = 4, 256, 224, 224
    M = B * HW
    x = rnd(M, K, seed=1).half()
    w = rnd(N, K, scale=1 / math.sqrt(K), seed=2)
    b = rnd(N, scale=0.1, seed=3)
    ref = x.float() @ w.half().float().t() + b
    wp, n_pad = ctx.pack_weight(w)
    bp = torch.zeros(n_pad, device=DEV); bp[:N] = b
    y = torch.zeros(M, N, device=DEV, dtype=torch.float16)
    ran = 0
    try:
        for cfg, name in as_cfgs(ctx):
            # GroupNorm partials
            d = lin_desc(x, wp, n_pad, N, y, bp)
            d.batch, d.in_h, d.in_w = B, HW, 1  # (rows of one sample are HW consecutive rows)
            sws = torch.zeros(ctx.gn_stats_floats(B, n_pad), device=DEV)
            d.gn_stats_ws, d.gn_groups = sws.data_ptr(), 32
            ctx.conv_override(cfg, 0)
            y.zero_()
            try:
                ctx.conv(d)
            except L.UpkError:
                continue
            mode, nblk = ctx.conv_gn_fused(d)
            torch.cuda.synchronize()
            check(y, ref)
            assert mode == 2, (name, mode)
            part = sws[: B * nblk * 2 * n_pad].reshape(B, nblk, 2, n_pad)
            yf = y.float().reshape(B, HW, N)
            check(part[:, :, 0, :N].sum(1), yf.sum(1), tol=2e-3)
            check(part[:, :, 1, :N].sum(1), (yf * yf).sum(1), tol=2e-3)
            # LayerNorm row sums
            d2 = lin_desc(x, wp, n_pad, N, y, bp)
            rows = torch.zeros(8, M, 2, device=DEV)
            d2.ln_rows_out = rows.data_ptr()
            slots = C_int_slots(ctx, d2)
            assert 1 <= slots <= 8, (name, slots)
            y.zero_()
            ctx.conv(d2)
            torch.cuda.synchronize()
            check(rows[:slots, :, 0].sum(0), y.float().sum(1), tol=2e-3)
            check(rows[:slots, :, 1].sum(0), (y.float() ** 2).sum(1), tol=2e-3)
            ran += 1
    finally:
        ctx.conv_override(-1, 0)
    assert ran >= 3


def C_int_slots(ctx, d):
    import ctypes as C
    s = C.c_int(0)
    ctx._chk(ctx.lib.upk_conv_ln_rows(ctx.h, C.byref(d), C.byref(s)))
    return s.value


def test_astat_refuses_what_it_cannot_run(ctx):
    """3x3 convs, K that is no multiple of the ring depth, tiles larger than the LDS: UPK_ESHAPE, never a wrong answer."""
    cfgs = as_cfgs(ctx)
    x = rnd(64, 96).half()   # 3 chunks: neither 7 | 3 nor 8 | 3
    w = rnd(32, 96)
    wp, n_pad = ctx.pack_weight(w)
    y = torch.zeros(64, 32, device=DEV, dtype=torch.float16)
    d = lin_desc(x, wp, n_pad, 32, y)
    try:
        for cfg, name in cfgs:
            ctx.conv_override(cfg, 0)
            with pytest.raises(L.UpkError):
                ctx.conv(d)
        xb = rnd(256, 7168).half()  # 224 chunks x 128 rows x 64 B = 1.8 MB > LDS
        wb, npb = ctx.pack_weight(rnd(64, 7168))
        yb = torch.zeros(256, 64, device=DEV, dtype=torch.float16)
        db = lin_desc(xb, wb, npb, 64, yb)
        ctx.conv_override(cfgs[0][0], 0)
        with pytest.raises(L.UpkError):
            ctx.conv(db)
    finally:
        ctx.conv_override(-1, 0)

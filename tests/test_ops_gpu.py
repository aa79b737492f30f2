"""Op-level parity of the HIP kernels (through the C ABI) against plain PyTorch fp32
references of the same op, fed the same fp16-rounded inputs."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from upgpt_amd import _lib as L

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def nhwc16(x):  # NCHW fp32 -> NHWC fp16 contiguous
    return x.permute(0, 2, 3, 1).contiguous().half()


def conv_ref(x, w, b, stride=1, ups=False, asym=False):
    x = x.half().float()
    w = w.half().float()
    if ups:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    if asym:
        x = F.pad(x, (0, 1, 0, 1))
        return F.conv2d(x, w, b, stride=2, padding=0)
    return F.conv2d(x, w, b, stride=stride, padding=w.shape[-1] // 2)


def make_desc(ctx, x1, w, bias, y, *, x2=None, stride=1, flags=0, n_out=None, residual=None, rowvec=None,
              rv_bs=0, rv_ss=0, step=None, H=None, W=None, row_map=None, col_map=None):
    B = x1.shape[0]
    H = H or x1.shape[1]
    W = W or x1.shape[2]
    wp, n_pad = ctx.pack_weight(w.contiguous(), row_map=row_map, col_map=col_map)
    d = L.ConvDesc()
    d.x1 = x1.data_ptr(); d.c1 = x1.shape[-1]; d.ld1 = x1.shape[-1]
    if x2 is not None:
        d.x2 = x2.data_ptr(); d.c2 = x2.shape[-1]; d.ld2 = x2.shape[-1]
    d.batch = B; d.in_h = H; d.in_w = W
    d.ksize = w.shape[-1] if w.dim() == 4 else 1
    d.stride = stride
    d.w_packed = wp.data_ptr(); d.n_pad = n_pad
    d.n_out = n_out if n_out is not None else w.shape[0]
    bp = None
    if bias is not None:
        bp = torch.zeros(n_pad, device=DEV)
        if row_map is not None:
            m = row_map.long()
            bp[: m.numel()] = torch.where(m >= 0, bias[m.clamp(min=0)], torch.zeros_like(bias[m.clamp(min=0)]))
        else:
            bp[: bias.numel()] = bias
        d.bias = bp.data_ptr()
    if residual is not None:
        d.residual = residual.data_ptr(); d.ld_res = residual.shape[-1]
    if rowvec is not None:
        d.rowvec = rowvec.data_ptr(); d.rv_batch_stride = rv_bs; d.rv_step_stride = rv_ss
    if step is not None:
        d.step = step.data_ptr()
    d.y = y.data_ptr(); d.ldy = y.shape[-1]
    d.flags = flags
    d._keep = (wp, bp, x1, x2, residual, rowvec, step, y)  # the descriptor holds raw pointers only
    return d


def check(got, ref, tol=2e-2):
    err = (got.float() - ref.float()).abs().max().item()
    scale = ref.float().abs().max().item() + 1e-6
    assert err / scale < tol, "max err %g vs scale %g" % (err, scale)


@pytest.mark.parametrize("cin,cout,hw,stride,ups", [
    (32, 224, (8, 6), 1, False), (224, 224, (32, 24), 1, False), (64, 96, (8, 8), 2, False),
    (448, 448, (16, 12), 2, False), (96, 64, (4, 3), 1, True), (896, 896, (4, 3), 1, False),
    (128, 16, (16, 16), 1, False),
])
def test_conv3x3(ctx, cin, cout, hw, stride, ups):
    B = 2
    H, W = hw
    x = rnd(B, cin, H, W)
    w = rnd(cout, cin, 3, 3, scale=1 / math.sqrt(9 * cin))
    b = rnd(cout, scale=0.1)
    ref = conv_ref(x, w, b, stride=stride, ups=ups)
    Ho, Wo = ref.shape[2:]
    y = torch.zeros(B, Ho, Wo, cout, device=DEV, dtype=torch.float16)
    d = make_desc(ctx, nhwc16(x), w, b, y, stride=stride, flags=L.F_UPSAMPLE2X if ups else 0)
    ctx.conv(d)
    torch.cuda.synchronize()
    check(y.permute(0, 3, 1, 2), ref)


def test_conv_all_configs_and_splitk(ctx):
    """Every compiled tile configuration and several split-K factors give the same answer."""
    B, cin, cout, H, W = 2, 64, 224, 12, 10
    x = rnd(B, cin, H, W)
    w = rnd(cout, cin, 3, 3, scale=1 / math.sqrt(9 * cin))
    b = rnd(cout, scale=0.1)
    ref = conv_ref(x, w, b)
    xn = nhwc16(x)
    ncfg = ctx.lib.upk_conv_num_configs()
    assert ncfg >= 8
    ran = 0
    try:
        for cfg in range(ncfg):
            name = ctx.lib.upk_conv_config_name(cfg).decode()
            for sk in (1, 2, 3):
                y = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
                ctx.conv_override(cfg, sk)
                try:
                    ctx.conv(make_desc(ctx, xn, w, b, y))
                except L.UpkError:
                    assert name.startswith("as"), name  # (only the A-stationary family refuses a 3x3: it is 1x1-only)
                    continue
                torch.cuda.synchronize()
                check(y.permute(0, 3, 1, 2), ref)
                ran += 1
    finally:
        ctx.conv_override(-1, 0)
    assert ran >= 3 * 62


def test_splitk_partials_that_cancel_and_exceed_fp16(ctx):
    """split-K partials leave as fp16 (igemm_common.h slab_store).  (a) Partials ~2000x the output that
    cancel across the K slices: the fp32 sum of the rounded slices keeps the output to the partials' 2^-11, i.e. the
    error is bounded by max|partial| 2^-10, not by inf / nan.  (b) Partials beyond 65504: the store saturates, the result
    stays finite (an fp32-slab build would be exact; the UNet / VAE never come near either regime)."""
    B, H, W, cin, cout = 1, 8, 8, 256, 64
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(B, cin, H, W, generator=g).to(DEV)
    w = (torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)).to(DEV)
    # first half of K: +big, second half: -big  ->  each split-K slice holds a partial of magnitude ~big * sqrt(cin / 2)
    big = 200.0
    x2 = x.clone()
    x2[:, : cin // 2] += big
    w2 = w.clone()
    w2[:, cin // 2:, 0, 0] = w[:, : cin // 2, 0, 0]  # same weights on both halves ...
    x2[:, cin // 2:] = x[:, cin // 2:] - (x2[:, : cin // 2] - x[:, : cin // 2])  # ... and the big term with opposite sign
    ref = conv_ref(x2, w2, None)
    part = F.conv2d(x2[:, : cin // 2].half().float(), w2[:, : cin // 2].half().float())
    assert float(part.abs().max()) > 50 * float(ref.abs().max())  # (the slices really are much larger than their sum)
    y = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
    try:
        ctx.conv_override(-1, 2)
        ctx.conv(make_desc(ctx, nhwc16(x2), w2, None, y))
        torch.cuda.synchronize()
    finally:
        ctx.conv_override(-1, 0)
    err = float((y.permute(0, 3, 1, 2).float() - ref).abs().max())
    assert torch.isfinite(y).all() and err <= float(part.abs().max()) * 2.0 ** -9, (err, float(part.abs().max()))
    # (b) beyond the fp16 range
    x3 = x2.clone()
    x3[:, : cin // 2] *= 400.0
    x3[:, cin // 2:] *= 400.0
    y3 = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
    try:
        ctx.conv_override(-1, 2)
        ctx.conv(make_desc(ctx, nhwc16(x3.clamp(-60000, 60000)), w2, None, y3))
        torch.cuda.synchronize()
    finally:
        ctx.conv_override(-1, 0)
    assert not torch.isnan(y3).any(), "saturated partials must not turn into nan"


@pytest.mark.parametrize("ks,c,c3,c4,cout,hw", [(3, 64, 96, 32, 224, (12, 10)), (3, 96, 64, 0, 64, (8, 8)),
                                               (1, 64, 32, 64, 96, (9, 7)), (3, 224, 448, 224, 224, (6, 4))])
def test_conv_with_appended_1x1_segment(ctx, ks, c, c3, c4, cout, hw):
    """y = conv_ks(h) + conv1x1(x3 | x4) + b1 + b2 in ONE launch (include/upk.h x3/x4: the ResBlock skip projection
    riding along the second conv, openaimodel.py:274-275) — every wave-specialised tile configuration x split-K,
    incl. splits that start inside the appended segment; the classic kernels refuse."""
    B = 2
    H, W = hw
    h = rnd(B, c, H, W, seed=1)
    x3 = rnd(B, c3, H, W, seed=2)
    x4 = rnd(B, c4, H, W, seed=3) if c4 else None
    w1 = rnd(cout, c, ks, ks, scale=1 / math.sqrt(ks * ks * c), seed=4)
    w2 = rnd(cout, c3 + c4, 1, 1, scale=1 / math.sqrt(c3 + c4), seed=5)
    b = rnd(cout, scale=0.1, seed=6)
    hq, x3q = h.half().float(), x3.half().float()
    xs = x3q if x4 is None else torch.cat([x3q, x4.half().float()], 1)
    ref = F.conv2d(hq, w1.half().float(), None, padding=ks // 2) + F.conv2d(xs, w2.half().float(), None) + b.view(1, -1, 1, 1)
    hn, x3n = nhwc16(h), nhwc16(x3)
    x4n = nhwc16(x4) if x4 is not None else None
    wp1, n_pad = ctx.pack_weight(w1.contiguous())
    wp2, n_pad2 = ctx.pack_weight(w2.contiguous())
    assert n_pad == n_pad2
    wp = torch.cat([wp1.reshape(-1), wp2.reshape(-1)])
    res = rnd(B, H, W, cout, seed=7).half()
    ncfg = ctx.lib.upk_conv_num_configs()
    ran = refused = 0
    try:
        for cfg in range(ncfg):
            for sk in (1, 2, 5):
                y = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
                d = make_desc(ctx, hn, w1, b, y, residual=res if cfg % 2 else None)
                d.w_packed = wp.data_ptr()
                d.x3, d.c3, d.ld3 = x3n.data_ptr(), c3, c3
                if x4n is not None:
                    d.x4, d.c4, d.ld4 = x4n.data_ptr(), c4, c4
                ctx.conv_override(cfg, sk)
                name = ctx.lib.upk_conv_config_name(cfg).decode()
                try:
                    ctx.conv(d)
                except RuntimeError:
                    refused += 1  # classic (register-staged) kernels, or a split finer than the K loop allows
                    continue
                torch.cuda.synchronize()
                want = ref + (res.float().permute(0, 3, 1, 2) if cfg % 2 else 0)
                check(y.permute(0, 3, 1, 2), want)
                assert "w" in name, "only the wave-specialised configurations take an appended segment (%s)" % name
                ran += 1
    finally:
        ctx.conv_override(-1, 0)
    assert ran >= 25 and refused >= 35, (ran, refused)  # (a 1x1 main conv has too few chunks for split-K)
    # cost-model choice (no override) must pick a configuration that supports it
    y = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
    d = make_desc(ctx, hn, w1, b, y)
    d.w_packed = wp.data_ptr()
    d.x3, d.c3, d.ld3 = x3n.data_ptr(), c3, c3
    if x4n is not None:
        d.x4, d.c4, d.ld4 = x4n.data_ptr(), c4, c4
    ctx.conv(d)
    torch.cuda.synchronize()
    check(y.permute(0, 3, 1, 2), ref)
    d.stride = 2
    with pytest.raises(RuntimeError):
        ctx.conv(d)


def test_conv_randomised_shapes_configs_and_epilogues(ctx):
    """Seeded sweep over ragged shapes (M not a tile multiple, N not a multiple of 16, 1x1 / 3x3, stride 2, upsample,
    concat), random tile configurations / split-K factors and epilogue combinations (bias, residual, timestep row
    vector, SiLU / quick-GELU, fp32 output, GroupNorm by-product) against F.conv2d.  Infeasible (config, split-K)
    pairs are refused by the library and skipped; everything that runs must be right."""
    g = torch.Generator(device="cpu").manual_seed(20240928)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    ncfg = ctx.lib.upk_conv_num_configs()
    ran = 0
    for case in range(60):
        ks = 3 if ri(0, 1) else 1
        stride = 2 if (ks == 3 and ri(0, 4) == 0) else 1
        ups = ks == 3 and stride == 1 and ri(0, 5) == 0
        B, H, W = ri(1, 3), ri(2, 13), ri(2, 13)
        c1 = 32 * ri(1, 6)
        c2 = 32 * ri(1, 3) if ri(0, 3) == 0 else 0
        cout = 4 * ri(1, 60)
        xa, xb = rnd(B, c1, H, W, seed=case), (rnd(B, c2, H, W, seed=1000 + case) if c2 else None)
        w = rnd(cout, c1 + c2, ks, ks, scale=1 / math.sqrt(ks * ks * (c1 + c2)), seed=2000 + case)
        b = rnd(cout, scale=0.1, seed=3000 + case) if ri(0, 3) else None
        ref = conv_ref(torch.cat([xa, xb], 1) if c2 else xa, w, b, stride=stride, ups=ups)
        Ho, Wo = ref.shape[2:]
        flags = L.F_UPSAMPLE2X if ups else 0
        rv = step = None
        if ri(0, 3) == 0:
            rv = rnd(3, B, cout, seed=4000 + case)
            step = torch.tensor([ri(0, 2)], dtype=torch.int32, device=DEV)
            ref = ref + rv[int(step)][:, :, None, None]
        act = ri(0, 5)
        if act == 0:
            flags |= L.F_SILU
            ref = F.silu(ref)
        elif act == 1:
            flags |= L.F_QUICKGELU
            ref = ref * torch.sigmoid(1.702 * ref)
        res = None
        if ri(0, 2) == 0:
            res = rnd(B, cout, Ho, Wo, seed=5000 + case)
            ref = ref + res.half().float()
        f32 = ri(0, 5) == 0
        y = torch.zeros(B, Ho, Wo, cout, device=DEV, dtype=torch.float32 if f32 else torch.float16)
        if f32:
            flags |= L.F_OUT_F32
        d = make_desc(ctx, nhwc16(xa), w, b, y, x2=nhwc16(xb) if c2 else None, stride=stride, flags=flags,
                      residual=nhwc16(res) if res is not None else None, rowvec=rv, rv_bs=cout, rv_ss=B * cout, step=step)
        sws = None
        if cout % 32 == 0 and ri(0, 1):
            sws = torch.zeros(ctx.gn_stats_floats(B, d.n_pad), device=DEV)
            d.gn_stats_ws, d.gn_groups = sws.data_ptr(), 32
        for trial in range(4):
            cfg, sk = (-1, 0) if trial == 0 else (ri(0, ncfg - 1), (1, 1, 2, 3, 4)[ri(0, 4)])
            ctx.conv_override(cfg, sk)
            y.zero_()
            try:
                ctx.conv(d)
            except L.UpkError:
                continue
            finally:
                ctx.conv_override(-1, 0)
            torch.cuda.synchronize()
            try:
                check(y.permute(0, 3, 1, 2), ref, tol=2e-2)
            except AssertionError as e:
                raise AssertionError("case %d: ks=%d stride=%d ups=%s B=%d %dx%d c1=%d c2=%d cout=%d flags=%#x bias=%s rv=%s res=%s "
                                     "f32=%s gn=%s cfg=%s sk=%d: %s" % (
                                         case, ks, stride, ups, B, H, W, c1, c2, cout, flags, b is not None, rv is not None,
                                         res is not None, f32, sws is not None,
                                         ctx.lib.upk_conv_config_name(cfg).decode() if cfg >= 0 else "auto", sk, e)) from None
            ran += 1
    assert ran >= 120


def test_conv_concat_rowvec_residual_silu(ctx):
    B, c1, c2, cout, H, W = 3, 64, 32, 96, 6, 5
    xa, xb = rnd(B, c1, H, W), rnd(B, c2, H, W, seed=1)
    w = rnd(cout, c1 + c2, 3, 3, scale=1 / math.sqrt(9 * (c1 + c2)))
    b = rnd(cout, scale=0.1)
    S = 4
    rv = rnd(S, B, cout, seed=2)
    step = torch.tensor([2], dtype=torch.int32, device=DEV)
    res = rnd(B, cout, H, W, seed=3)
    ref = conv_ref(torch.cat([xa, xb], 1), w, b) + rv[2][:, :, None, None]
    ref = F.silu(ref) + res.half().float()
    y = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
    d = make_desc(ctx, nhwc16(xa), w, b, y, x2=nhwc16(xb), flags=L.F_SILU, residual=nhwc16(res), rowvec=rv,
                  rv_bs=cout, rv_ss=B * cout, step=step)
    ctx.conv(d)
    torch.cuda.synchronize()
    check(y.permute(0, 3, 1, 2), ref)


def test_conv_asym_pad_and_nchw_out(ctx):
    B, cin, cout, H, W = 2, 32, 4, 8, 6
    x = rnd(B, cin, H, W)
    w = rnd(cout, cin, 3, 3, scale=0.1)
    b = rnd(cout, scale=0.1)
    ref = conv_ref(x, w, b, asym=True)
    y = torch.zeros(B, ref.shape[2], ref.shape[3], 16, device=DEV, dtype=torch.float16)
    ctx.conv(make_desc(ctx, nhwc16(x), w, b, y, stride=2, flags=L.F_PAD_ASYM))
    torch.cuda.synchronize()
    check(y[..., :cout].permute(0, 3, 1, 2), ref)
    assert (y[..., cout:] == 0).all()
    # fp32 NCHW output straight from the epilogue (UNet 'out' conv, VAE conv_out)
    ref2 = conv_ref(x, w, b)
    y2 = torch.zeros(B, cout, H, W, device=DEV)
    d = make_desc(ctx, nhwc16(x), w, b, y2, flags=L.F_OUT_NCHW_F32)
    d.ldy = 0
    ctx.conv(d)
    torch.cuda.synchronize()
    check(y2, ref2, tol=2e-3)


@pytest.mark.parametrize("M,K,N", [(96, 896, 224), (6144, 224, 224), (17, 768, 256), (50, 224, 896), (4608, 448, 448)])
def test_gemm_bias_residual(ctx, M, K, N):
    a = rnd(M, K).half()
    w = rnd(N, K, scale=1 / math.sqrt(K))
    b = rnd(N, scale=0.1)
    res = rnd(M, N, seed=5).half()
    ref = a.float() @ w.half().float().t() + b + res.float()
    wp, n_pad = ctx.pack_weight(w)
    bp = torch.zeros(n_pad, device=DEV); bp[:N] = b
    y = torch.zeros(M, N, device=DEV, dtype=torch.float16)
    ctx.gemm(a, K, M, K, wp, N, n_pad, bp, res, N, y, N, 0)
    torch.cuda.synchronize()
    check(y, ref)
    # fp32 output
    y32 = torch.zeros(M, N, device=DEV)
    ctx.gemm(a, K, M, K, wp, N, n_pad, bp, None, 0, y32, N, L.F_OUT_F32)
    torch.cuda.synchronize()
    check(y32, ref - res.float(), tol=3e-3)


def geglu_row_map(n_half):
    """packed rows: per 64-row block [32 value rows | 32 gate rows] (include/upk.h UPK_F_GEGLU)."""
    idx = torch.arange(2 * n_half)
    blk, j = idx // 64, idx % 64
    return torch.where(j < 32, blk * 32 + j, n_half + blk * 32 + (j - 32)).int()


@pytest.mark.parametrize("M,d", [(768, 224), (96, 896)])
def test_gemm_geglu(ctx, M, d):
    inner = 4 * d
    a = rnd(M, d).half()
    w = rnd(2 * inner, d, scale=1 / math.sqrt(d))
    b = rnd(2 * inner, scale=0.1)
    h = a.float() @ w.half().float().t() + b
    ref = h[:, :inner] * F.gelu(h[:, inner:])
    rm = geglu_row_map(inner).to(DEV)
    wp, n_pad = ctx.pack_weight(w, row_map=rm)
    bp = b[rm.long()].contiguous()
    y = torch.zeros(M, inner, device=DEV, dtype=torch.float16)
    for sk in ((0, 2) if d >= 512 else (0,)):
        ctx.conv_override(-1, sk)
        y.zero_()
        ctx.gemm(a, d, M, d, wp, inner, n_pad, bp, None, 0, y, inner, L.F_GEGLU)
        torch.cuda.synchronize()
        check(y, ref)
    ctx.conv_override(-1, 0)


@pytest.mark.parametrize("M,d,N,geglu", [(8192, 224, 768, False), (600, 448, 512, False), (96, 896, 896, False),
                                          (512, 224, 1792, True)])
def test_gemm_with_folded_layernorm(ctx, M, d, N, geglu):
    """LayerNorm folded into its consumer Linear (upk_conv_desc.ln_colsum): rows are the raw residual
    stream (offset mean, per-row scale), weights W*gamma, bias b + W@beta (attention.py:203-215)."""
    g = torch.Generator(device="cpu").manual_seed(11)
    x = (torch.randn(M, d, generator=g) * (0.5 + 3 * torch.rand(M, 1, generator=g)) + 4 * torch.randn(M, 1, generator=g))
    x = x.to(DEV).half()
    gamma = 1 + 0.2 * rnd(d, seed=2)
    beta = 0.1 * rnd(d, seed=3)
    w = rnd(N, d, scale=1 / math.sqrt(d), seed=4)
    b = rnd(N, scale=0.1, seed=5)
    h = F.layer_norm(x.float(), (d,), gamma, beta, 1e-5) @ w.t() + b
    if geglu:
        inner = N // 2
        ref = h[:, :inner] * F.gelu(h[:, inner:])
        rm = geglu_row_map(inner).to(DEV)
    else:
        ref, rm = h, None
    wf = (w * gamma[None, :]).contiguous()
    wp, n_pad = ctx.pack_weight(wf, row_map=rm)
    bf = b + w @ beta
    u = wf.half().float().sum(dim=1)
    if rm is not None:
        bf, u = bf[rm.long()], u[rm.long()]
    bp = torch.zeros(n_pad, device=DEV); bp[: bf.numel()] = bf
    up = torch.zeros(n_pad, device=DEV); up[: u.numel()] = u
    n_out = N // 2 if geglu else N
    y = torch.zeros(M, n_out, device=DEV, dtype=torch.float16)
    dsc = L.ConvDesc()
    dsc.x1 = x.data_ptr(); dsc.c1 = d; dsc.ld1 = d; dsc.batch = 1; dsc.in_h = M; dsc.in_w = 1
    dsc.ksize = 1; dsc.stride = 1; dsc.w_packed = wp.data_ptr(); dsc.n_out = n_out; dsc.n_pad = n_pad
    dsc.bias = bp.data_ptr(); dsc.y = y.data_ptr(); dsc.ldy = n_out; dsc.flags = L.F_GEGLU if geglu else 0
    dsc.ln_colsum = up.data_ptr(); dsc.ln_eps = 1e-5; dsc.ln_dim = d
    ctx.conv(dsc)
    torch.cuda.synchronize()
    check(y, ref, tol=4e-3)
    # every configuration that supports the fold agrees; the others are refused, never silently wrong
    ok = 0
    for cfg in range(ctx.lib.upk_conv_num_configs()):
        ctx.conv_override(cfg, 1)
        y.zero_()
        try:
            ctx.conv(dsc)
        except L.UpkError:
            continue
        finally:
            ctx.conv_override(-1, 0)
        torch.cuda.synchronize()
        check(y, ref, tol=4e-3)
        ok += 1
    assert ok >= 4
    ctx.conv_override(-1, 2)
    with pytest.raises(L.UpkError):
        ctx.conv(dsc)
    ctx.conv_override(-1, 0)


@pytest.mark.parametrize("cin,cout,hw,res", [(896, 896, (8, 8), True), (448, 448, (16, 12), False), (1792, 896, (4, 4), True)])
def test_splitk_reduce_emits_groupnorm_partials(ctx, cin, cout, hw, res):
    """A split-K conv's reduce pass writes the GroupNorm partial sums of its output (upk_conv_desc.gn_stats_ws):
    GroupNorm apply-only on them must equal the two-pass GroupNorm bit for bit (fp16 rounding where the full GroupNorm is
    the one-launch kernel for small feature maps); without split-K nothing is produced and upk_conv_gn_fused says so."""
    B, (H, W) = 3, hw
    x = rnd(B * H * W, cin).half()
    w = rnd(cout, cin, 3, 3, scale=1 / math.sqrt(9 * cin))
    b = rnd(cout, scale=0.1)
    r = rnd(B * H * W, cout, seed=7).half() if res else None
    wp, n_pad = ctx.pack_weight(w)
    bp = torch.zeros(n_pad, device=DEV); bp[:cout] = b
    y = torch.zeros(B * H * W, cout, device=DEV, dtype=torch.float16)
    sws = torch.full((ctx.gn_stats_floats(B, n_pad),), float("nan"), device=DEV)
    d = L.ConvDesc()
    d.x1 = x.data_ptr(); d.c1 = cin; d.ld1 = cin; d.batch = B; d.in_h = H; d.in_w = W; d.ksize = 3; d.stride = 1
    d.w_packed = wp.data_ptr(); d.n_out = cout; d.n_pad = n_pad; d.bias = bp.data_ptr(); d.y = y.data_ptr(); d.ldy = cout
    if res:
        d.residual = r.data_ptr(); d.ld_res = cout
    d.gn_stats_ws = sws.data_ptr(); d.gn_groups = 32
    ref = F.conv2d(x.float().view(B, H, W, cin).permute(0, 3, 1, 2), w.half().float(), b, padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(B * H * W, cout) + (r.float() if res else 0)
    gamma, beta = 1 + 0.1 * rnd(cout, seed=2), 0.1 * rnd(cout, seed=3)
    ws2 = torch.zeros(ctx.groupnorm_ws_bytes(B, H * W) // 4, device=DEV)
    for sk in (4, 9):
        ctx.conv_override(-1, sk)
        assert ctx.conv_gn_fused(d)[0] == 1
        y.zero_(); sws.fill_(float("nan"))
        ctx.conv(d)
        ctx.conv_override(-1, 0)
        torch.cuda.synchronize()
        check(y, ref)
        full = torch.zeros_like(y); app = torch.zeros_like(y)
        ctx.groupnorm(y, cout, cout, None, 0, 0, B, H * W, 32, gamma, beta, 1e-5, True, full, cout, ws2)
        ctx._chk(ctx.lib.upk_groupnorm_apply_nhwc_f16(ctx.h, y.data_ptr(), cout, cout, None, 0, 0, B, H * W, 32,
                                                      gamma.data_ptr(), beta.data_ptr(), 1e-5, 1, app.data_ptr(), cout,
                                                      sws.data_ptr(), 1, 0, n_pad, None, 0, 0, ctx._s()))
        torch.cuda.synchronize()
        if H * W > 64:
            assert torch.equal(full, app)
        else:  # (small feature maps: upk_groupnorm_nhwc_f16 is the one-launch kernel, its sums run in another order)
            assert (app.float() - full.float()).abs().max().item() <= 2e-3 * full.float().abs().max().item()
    # without split-K: per-(M tile, channel) partials from the epilogue (mode 2) where the tile configuration allows
    # it (M tiles inside one sample, not the K-split kernels), otherwise mode 0 and nothing is promised
    gref = F.silu(F.group_norm(ref.view(B, H * W, cout).permute(0, 2, 1), 32, gamma, beta, 1e-5)).permute(0, 2, 1)
    modes = set()
    for cfg in range(ctx.lib.upk_conv_num_configs()):
        ctx.conv_override(cfg, 1)
        try:
            mode, nblk = ctx.conv_gn_fused(d)
            modes.add(mode)
            assert mode in (0, 2)
            if mode == 2:
                y.zero_(); sws.fill_(float("nan"))
                ctx.conv(d)
                app = torch.zeros_like(y)
                ctx._chk(ctx.lib.upk_groupnorm_apply_nhwc_f16(
                    ctx.h, y.data_ptr(), cout, cout, None, 0, 0, B, H * W, 32, gamma.data_ptr(), beta.data_ptr(), 1e-5, 1,
                    app.data_ptr(), cout, sws.data_ptr(), 2, nblk, n_pad, None, 0, 0, ctx._s()))
                torch.cuda.synchronize()
                check(y, ref)
                full = torch.zeros_like(y)
                ctx.groupnorm(y, cout, cout, None, 0, 0, B, H * W, 32, gamma, beta, 1e-5, True, full, cout, ws2)
                torch.cuda.synchronize()
                assert (app.float() - full.float()).abs().max().item() <= 2e-3 * full.float().abs().max().item()
                check(app.view(B, H * W, cout), gref, tol=1e-2)
        except L.UpkError:
            pass
        finally:
            ctx.conv_override(-1, 0)
    assert 2 in modes


def test_groupnorm_apply_from_two_producers_channel_partials(ctx):
    """Decoder ResBlock input: GroupNorm over the concat [h, skip] whose two sources were produced by different unsplit
    conv launches; both leave per-(M tile, channel) partials and the GroupNorm runs its apply pass only."""
    B, H, W = 2, 16, 16
    outs, descs, bufs = [], [], []
    for (cin, cout, seed) in ((64, 448, 1), (96, 224, 2)):
        x = rnd(B * H * W, cin, seed=seed).half()
        w = rnd(cout, cin, 3, 3, scale=1 / math.sqrt(9 * cin), seed=10 + seed)
        wp, n_pad = ctx.pack_weight(w)
        y = torch.zeros(B * H * W, cout, device=DEV, dtype=torch.float16)
        sws = torch.zeros(ctx.gn_stats_floats(B, n_pad), device=DEV)
        d = L.ConvDesc()
        d.x1 = x.data_ptr(); d.c1 = cin; d.ld1 = cin; d.batch = B; d.in_h = H; d.in_w = W; d.ksize = 3; d.stride = 1
        d.w_packed = wp.data_ptr(); d.n_out = cout; d.n_pad = n_pad; d.y = y.data_ptr(); d.ldy = cout
        d.gn_stats_ws = sws.data_ptr(); d.gn_groups = 32
        ctx.conv_override(-1, 1)
        mode, nblk = ctx.conv_gn_fused(d)
        ctx.conv(d)
        ctx.conv_override(-1, 0)
        assert mode == 2
        outs.append(y); descs.append((sws, nblk, n_pad)); bufs.append((x, wp, d))
    torch.cuda.synchronize()
    h, sk = outs
    C = h.shape[1] + sk.shape[1]
    gamma, beta = 1 + 0.1 * rnd(C, seed=5), 0.1 * rnd(C, seed=6)
    full, app = torch.zeros(B * H * W, C, device=DEV, dtype=torch.float16), torch.zeros(B * H * W, C, device=DEV, dtype=torch.float16)
    ws = torch.zeros(ctx.groupnorm_ws_bytes(B, H * W) // 4, device=DEV)
    ctx.groupnorm(h, h.shape[1], h.shape[1], sk, sk.shape[1], sk.shape[1], B, H * W, 32, gamma, beta, 1e-5, True, full, C, ws)
    ctx._chk(ctx.lib.upk_groupnorm_apply_nhwc_f16(
        ctx.h, h.data_ptr(), h.shape[1], h.shape[1], sk.data_ptr(), sk.shape[1], sk.shape[1], B, H * W, 32,
        gamma.data_ptr(), beta.data_ptr(), 1e-5, 1, app.data_ptr(), C, descs[0][0].data_ptr(), 2, descs[0][1], descs[0][2],
        descs[1][0].data_ptr(), descs[1][1], descs[1][2], ctx._s()))
    torch.cuda.synchronize()
    assert (app.float() - full.float()).abs().max().item() <= 2e-3 * full.float().abs().max().item()
    x = torch.cat([h, sk], 1).float().view(B, H * W, C)
    ref = F.silu(F.group_norm(x.permute(0, 2, 1), 32, gamma, beta, 1e-5)).permute(0, 2, 1)
    check(app.view(B, H * W, C), ref, tol=1e-2)
    with pytest.raises(L.UpkError):  # per-group partials cannot describe a concat
        ctx._chk(ctx.lib.upk_groupnorm_apply_nhwc_f16(
            ctx.h, h.data_ptr(), h.shape[1], h.shape[1], sk.data_ptr(), sk.shape[1], sk.shape[1], B, H * W, 32,
            gamma.data_ptr(), beta.data_ptr(), 1e-5, 1, app.data_ptr(), C, descs[0][0].data_ptr(), 1, 0, descs[0][2],
            None, 0, 0, ctx._s()))


def test_qkv_gemm_with_transposed_v(ctx):
    """Fused q|k|v projection: q,k token-major, v written as V^T [B, heads, dpad, vt_ld];
    head dim 28 padded to 32 by the packing row map."""
    B, n, dm, heads, dh, dp = 2, 48, 224, 8, 28, 32
    x = rnd(B * n, dm).half()
    wq, wk, wv = (rnd(heads * dh, dm, scale=1 / math.sqrt(dm), seed=s) for s in (1, 2, 3))
    w = torch.cat([wq, wk, wv], 0).contiguous()
    hd = heads * dp
    j = torch.arange(3 * hd)
    part, r = j // hd, j % hd
    h_, d_ = r // dp, r % dp
    rm = torch.where(d_ < dh, part * heads * dh + h_ * dh + d_, torch.full_like(j, -1)).int().to(DEV)
    wp, n_pad = ctx.pack_weight(w, row_map=rm)
    assert n_pad == 3 * hd
    vt_ld = 64
    qk = torch.zeros(B * n, 2 * hd, device=DEV, dtype=torch.float16)
    vt = torch.zeros(B, heads, dp, vt_ld, device=DEV, dtype=torch.float16)
    d = L.ConvDesc()
    d.x1 = x.data_ptr(); d.c1 = dm; d.ld1 = dm; d.batch = 1; d.in_h = B * n; d.in_w = 1; d.ksize = 1; d.stride = 1
    d.w_packed = wp.data_ptr(); d.n_pad = n_pad; d.n_out = 2 * hd
    d.y = qk.data_ptr(); d.ldy = 2 * hd
    d.vt = vt.data_ptr(); d.vt_from = 2 * hd; d.vt_heads = heads; d.vt_dhead = dp; d.vt_ld = vt_ld; d.vt_tokens = n
    ctx.conv(d)
    torch.cuda.synchronize()
    xf = x.float()
    for name, wmat, got in (("q", wq, qk[:, :hd]), ("k", wk, qk[:, hd:])):
        ref = (xf @ wmat.half().float().t()).view(B * n, heads, dh)
        g = got.view(B * n, heads, dp)
        check(g[..., :dh], ref)
        assert (g[..., dh:] == 0).all()
    vref = (xf @ wv.half().float().t()).view(B, n, heads, dh).permute(0, 2, 3, 1)  # B,h,d,n
    check(vt[:, :, :dh, :n], vref)
    assert (vt[:, :, dh:, :] == 0).all() and (vt[..., n:] == 0).all()


@pytest.mark.parametrize("d,nq,nkv,heads", [(32, 768, 768, 8), (64, 192, 87, 8), (128, 48, 48, 8), (128, 12, 12, 8),
                                            (32, 100, 87, 2), (512, 200, 200, 1),
                                            # d >= 256 (VAE mid-block): 1 / 2 / 4 waves per workgroup by launch size
                                            (512, 1024, 1024, 1), (256, 600, 130, 16), (256, 1024, 96, 16),
                                            # LDS-staged K/V tiles (nkv % 64 == 0, >= 256): QT=2 and QT=1, ragged nq
                                            (32, 1024, 1024, 32), (32, 1000, 256, 3), (64, 256, 256, 8), (64, 200, 320, 40)])
def test_attention(ctx, d, nq, nkv, heads):
    B = 2
    q = rnd(B, nq, heads * d).half()
    k = rnd(B, nkv, heads * d, seed=1).half()
    v = rnd(B, nkv, heads * d, seed=2).half()
    scale = d ** -0.5
    vt_ld = (nkv + 31) // 32 * 32
    vt = torch.zeros(B, heads, d, vt_ld, device=DEV, dtype=torch.float16)
    vt[..., :nkv] = v.view(B, nkv, heads, d).permute(0, 2, 3, 1)
    out = torch.zeros(B, nq, heads * d, device=DEV, dtype=torch.float16)
    ctx.attention(q, heads * d, nq * heads * d, k, heads * d, nkv * heads * d, vt, vt_ld, out, heads * d,
                  nq * heads * d, B, heads, nq, nkv, d, scale)
    torch.cuda.synchronize()
    qf = q.float().view(B, nq, heads, d).transpose(1, 2)
    kf = k.float().view(B, nkv, heads, d).transpose(1, 2)
    vf = v.float().view(B, nkv, heads, d).transpose(1, 2)
    ref = torch.softmax(qf @ kf.transpose(-1, -2) * scale, -1) @ vf
    check(out.view(B, nq, heads, d).transpose(1, 2), ref, tol=1e-2)


def test_attention_and_norms_randomised(ctx):
    """Seeded sweep: attention over ragged (n_q, n_kv) incl. the LDS-staged and the causal forms; GroupNorm over odd
    pixel counts / two sources / group widths that straddle 16-byte vectors; LayerNorm over every supported width."""
    g = torch.Generator(device="cpu").manual_seed(7)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    for case in range(24):
        d, heads, B = (32, 64, 128)[ri(0, 2)], ri(1, 4), ri(1, 3)
        nq = ri(1, 300)
        nkv = nq if ri(0, 2) == 0 else (64 * ri(4, 6) if ri(0, 2) == 0 else ri(1, 300))
        causal = nq == nkv and ri(0, 1) == 1
        q = rnd(B, nq, heads * d, seed=case).half()
        k = rnd(B, nkv, heads * d, seed=100 + case).half()
        v = rnd(B, nkv, heads * d, seed=200 + case).half()
        vt_ld = (nkv + 31) // 32 * 32
        vt = torch.zeros(B, heads, d, vt_ld, device=DEV, dtype=torch.float16)
        vt[..., :nkv] = v.view(B, nkv, heads, d).permute(0, 2, 3, 1)
        out = torch.zeros(B, nq, heads * d, device=DEV, dtype=torch.float16)
        a = (q, heads * d, nq * heads * d, k, heads * d, nkv * heads * d, vt, vt_ld, out, heads * d, nq * heads * d, B, heads)
        if causal:
            ctx.attention_causal(*a, nq, d, d ** -0.5)
        else:
            ctx.attention(*a, nq, nkv, d, d ** -0.5)
        torch.cuda.synchronize()
        qf = q.float().view(B, nq, heads, d).transpose(1, 2)
        kf = k.float().view(B, nkv, heads, d).transpose(1, 2)
        vf = v.float().view(B, nkv, heads, d).transpose(1, 2)
        sc = qf @ kf.transpose(-1, -2) * d ** -0.5
        if causal:
            sc = sc.masked_fill(torch.ones(nq, nkv, device=DEV, dtype=torch.bool).triu(1), float("-inf"))
        ref = torch.softmax(sc, -1) @ vf
        got = out.view(B, nq, heads, d).transpose(1, 2).float()
        assert torch.isfinite(got).all(), (case, d, nq, nkv, causal)
        assert (got - ref).abs().max().item() < 1e-2 * ref.abs().max().item(), (case, d, nq, nkv, causal)
    for case in range(16):
        C = 32 * ri(1, 24)
        c2 = 8 * ri(0, C // 8 - 1) if ri(0, 1) else 0
        c1 = C - c2
        if c1 % 8:
            c1, c2 = C, 0
        B, hw, silu = ri(1, 3), ri(1, 700), bool(ri(0, 1))
        xa = (rnd(B, hw, c1, seed=300 + case) * 2 + 0.5).half()
        xb = (rnd(B, hw, c2, seed=400 + case) - 0.3).half() if c2 else None
        gamma, beta = 1 + 0.1 * rnd(C, seed=500 + case), 0.1 * rnd(C, seed=600 + case)
        x = torch.cat([xa, xb], -1) if c2 else xa
        ref = F.group_norm(x.float().permute(0, 2, 1), 32, gamma, beta, 1e-5).permute(0, 2, 1)
        ref = F.silu(ref) if silu else ref
        y = torch.zeros(B, hw, C, device=DEV, dtype=torch.float16)
        ws = torch.zeros(ctx.groupnorm_ws_bytes(B, hw) // 4, device=DEV)
        ctx.groupnorm(xa, c1, c1, xb, c2, c2, B, hw, 32, gamma, beta, 1e-5, silu, y, C, ws)
        torch.cuda.synchronize()
        assert (y.float() - ref).abs().max().item() < 1e-2 * max(1.0, ref.abs().max().item()), (case, c1, c2, hw)
    for case in range(12):
        dln, rows = 8 * ri(1, 256), ri(1, 400)
        x = (rnd(rows, dln, seed=700 + case) * 3 + 1).half()
        gamma, beta = 1 + 0.1 * rnd(dln, seed=800 + case), 0.1 * rnd(dln, seed=900 + case)
        y = torch.zeros(rows, dln, device=DEV, dtype=torch.float16)
        ctx.layernorm(x, dln, rows, dln, gamma, beta, 1e-5, y, dln)
        torch.cuda.synchronize()
        ref = F.layer_norm(x.float(), (dln,), gamma, beta, 1e-5)
        assert (y.float() - ref).abs().max().item() < 1e-2 * max(1.0, ref.abs().max().item()), (case, dln, rows)


def test_attention_online_softmax_rescale(ctx):
    """A late key with a huge score forces the running-max rescale branch."""
    B, heads, d, nq, nkv = 1, 1, 64, 16, 96
    q = rnd(B, nq, d).half()
    k = rnd(B, nkv, d, seed=1).half()
    k[0, 70] = q[0, 3] * 4
    v = rnd(B, nkv, d, seed=2).half()
    vt = v.permute(0, 2, 1).contiguous().view(B, 1, d, nkv)
    out = torch.zeros(B, nq, d, device=DEV, dtype=torch.float16)
    ctx.attention(q, d, nq * d, k, d, nkv * d, vt, nkv, out, d, nq * d, B, heads, nq, nkv, d, 1.0)
    torch.cuda.synchronize()
    ref = torch.softmax(q.float() @ k.float().transpose(-1, -2), -1) @ v.float()
    check(out, ref, tol=1e-2)


@pytest.mark.parametrize("c1,c2,hw,silu,eps", [(224, 0, 768, True, 1e-5), (896, 448, 192, True, 1e-5),
                                               (448, 0, 12, False, 1e-6), (128, 0, 4096, True, 1e-6),
                                               (896, 896, 48, True, 1e-5), (448, 224, 768, True, 1e-5),
                                               # small feature maps: one launch (gn_onepass_kernel) — 8-, 4- and 2-wide
                                               # vectors, groups straddling the concat seam, one value per thread, and
                                               # odd group widths (two-pass fallback)
                                               (896, 896, 16, True, 1e-5), (896, 448, 64, True, 1e-5),
                                               (896, 0, 64, False, 1e-5), (448, 224, 16, True, 1e-6),
                                               (1024, 1024, 64, True, 1e-5), (32, 0, 4, True, 1e-5), (96, 64, 64, False, 1e-5)])
def test_groupnorm(ctx, c1, c2, hw, silu, eps):
    B, C = 3, c1 + c2
    xa = (rnd(B, hw, c1) * 2 + 0.5).half()
    xb = (rnd(B, hw, c2, seed=1) - 0.3).half() if c2 else None
    gamma = 1 + 0.1 * rnd(C, seed=2)
    beta = 0.1 * rnd(C, seed=3)
    x = torch.cat([xa, xb], -1) if c2 else xa
    ref = F.group_norm(x.float().permute(0, 2, 1), 32, gamma, beta, eps).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    y = torch.zeros(B, hw, C, device=DEV, dtype=torch.float16)
    ws = torch.zeros(ctx.groupnorm_ws_bytes(B, hw) // 4, device=DEV)
    ctx.groupnorm(xa, c1, c1, xb, c2, c2 if c2 else 0, B, hw, 32, gamma, beta, eps, silu, y, C, ws)
    torch.cuda.synchronize()
    check(y, ref, tol=4e-3)
    y2 = torch.zeros_like(y)
    ctx.groupnorm(xa, c1, c1, xb, c2, c2 if c2 else 0, B, hw, 32, gamma, beta, eps, silu, y2, C, ws)
    torch.cuda.synchronize()
    assert torch.equal(y, y2), "GroupNorm must be bitwise reproducible"


@pytest.mark.parametrize("rows,d", [(6144, 224), (1536, 448), (97, 896), (5, 1024), (33, 2048)])
def test_layernorm(ctx, rows, d):
    x = (rnd(rows, d) * 3 + 1).half()
    gamma = 1 + 0.1 * rnd(d, seed=2)
    beta = 0.1 * rnd(d, seed=3)
    ref = F.layer_norm(x.float(), (d,), gamma, beta, 1e-5)
    y = torch.zeros(rows, d, device=DEV, dtype=torch.float16)
    ctx.layernorm(x, d, rows, d, gamma, beta, 1e-5, y, d)
    torch.cuda.synchronize()
    check(y, ref, tol=2e-3)


def test_timestep_embed(ctx):
    t = torch.tensor([0., 1., 21., 500., 981., 999.], device=DEV)
    dim = 224
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half).to(DEV)
    args = t[:, None] * freqs[None]
    ref = torch.cat([torch.cos(args), torch.sin(args)], -1)
    out = torch.zeros(6, 224, device=DEV, dtype=torch.float16)
    ctx.timestep_embed(t, 6, dim, 10000.0, out, 224)
    torch.cuda.synchronize()
    assert (out.float() - ref).abs().max() < 2e-3


def test_layout_and_ddim_step(ctx):
    B, Cc, H, W = 2, 4, 6, 5
    hw = H * W
    x = rnd(B, Cc, H, W)
    mask = rnd(B, 1, H, W, seed=1)
    xin = torch.full((B, hw, 32), 7.0, device=DEV, dtype=torch.float16)
    ctx.nchw_to_nhwc(x, B, Cc, hw, xin, 32, 0, 0, 1.0)
    ctx.nchw_to_nhwc(mask, B, 1, hw, xin, 32, Cc, 32, 1.0)
    torch.cuda.synchronize()
    ref = torch.cat([x, mask], 1).permute(0, 2, 3, 1).reshape(B, hw, 5).half()
    assert torch.equal(xin[..., :5], ref) and (xin[..., 5:] == 0).all()
    back = torch.zeros(B, 5, H, W, device=DEV)
    ctx.nhwc_to_nchw(xin, 32, B, 5, hw, back)
    torch.cuda.synchronize()
    assert torch.equal(back, torch.cat([x, mask], 1).half().float())
    # ddim update (ddim.py:189-203)
    S = 3
    coefs = torch.rand(S, 4, device=DEV) + 0.1
    noise = rnd(S, B, Cc, H, W, seed=4)
    e = rnd(B, Cc, H, W, seed=5)
    step = torch.tensor([1], dtype=torch.int32, device=DEV)
    x0 = x.clone()
    pred = torch.zeros_like(x)
    ctx.ddim_step(x, e, coefs, noise, step, pred, xin, 32, B, Cc, hw)
    ctx.advance_step(step)
    torch.cuda.synchronize()
    c = coefs[1]
    p_ref = (x0 - c[0] * e) * c[1]
    x_ref = c[2] * p_ref + c[3] * e + noise[1]
    assert torch.allclose(pred, p_ref, atol=1e-5) and torch.allclose(x, x_ref, atol=1e-5)
    assert torch.equal(xin[..., :4], x.permute(0, 2, 3, 1).reshape(B, hw, 4).half())
    assert step.item() == 2


def test_graph_capture_replay(ctx):
    """A captured sequence replays with the device-side step counter advancing."""
    M, K, N = 64, 64, 64
    a = rnd(M, K).half()
    w = rnd(N, K, scale=0.1)
    wp, n_pad = ctx.pack_weight(w)
    y = torch.zeros(M, N, device=DEV, dtype=torch.float16)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        ctx.graph_begin()
        ctx.gemm(a, K, M, K, wp, N, n_pad, None, None, 0, y, N, 0)
        ctx.advance_step(step)
        g = ctx.graph_end()
        for _ in range(5):
            ctx.graph_launch(g)
    s.synchronize()
    assert step.item() == 5
    check(y, a.float() @ w.half().float().t())
    ctx.graph_destroy(g)


def test_errors_are_reported(ctx):
    d = L.ConvDesc()
    with pytest.raises(L.UpkError):
        ctx.conv(d)
    a = rnd(16, 40).half()
    with pytest.raises(L.UpkError) as ei:
        ctx.gemm(a, 40, 16, 40, a, 16, 16, None, None, 0, a, 16, 0)  # K not a multiple of 32
    assert ei.value.code == -2


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,cin,cout,ks,sk,res,rv,silu,skip_y", [
    (4, 4, 896, 896, 3, 9, False, True, True, True),     # level-3 ResBlock conv1 -> out_layers norm (cpg 28, 1 vector/thread)
    (8, 8, 896, 896, 3, 9, True, False, True, False),    # level-2 conv2 + residual -> next block's norm
    (8, 6, 448, 896, 3, 4, False, True, True, False),    # 256x192 level 2
    (8, 8, 448, 896, 1, 2, True, False, False, False),    # SpatialTransformer.norm behind a split 1x1 (no SiLU)
    (16, 16, 448, 448, 3, 4, True, False, False, False),  # cpg 14 / 7 / 2: outside the fused pass (it measured slower
    (32, 32, 224, 224, 3, 4, False, False, True, False),  # there): plain reduce, request ignored and reported so
    (16, 12, 256, 64, 1, 2, True, True, True, False),
])
def test_splitk_reduce_applies_groupnorm(ctx, H, W, cin, cout, ks, sk, res, rv, silu, skip_y):
    """include/upk.h gno_*: the reduce pass of a split-K conv writes y AND SiLU?(GroupNorm32(y)); a launch that does not
    split K ignores the request and reports so."""
    B = 2
    x = rnd(B, cin, H, W)
    w = rnd(cout, cin, ks, ks, scale=1 / math.sqrt(ks * ks * cin))
    b = rnd(cout, scale=0.1)
    gamma, beta = rnd(cout, seed=3) * 0.5 + 1.0, rnd(cout, seed=4) * 0.3
    resid = rnd(B, H, W, cout, seed=7).half() if res else None
    rowv = rnd(3, B, cout, seed=8) if rv else None  # [step][sample][channel], step counter on the device
    step = torch.tensor([2], device=DEV, dtype=torch.int32) if rv else None
    ref = F.conv2d(x.half().float(), w.half().float(), b, padding=ks // 2)
    if res:
        ref = ref + resid.float().permute(0, 3, 1, 2)
    if rv:
        ref = ref + rowv[2].view(B, cout, 1, 1)
    ref_h = ref.half().float()
    ref_n = F.group_norm(ref_h, 32, gamma, beta, 1e-5)
    if silu:
        ref_n = F.silu(ref_n)
    y = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
    yn = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
    d = make_desc(ctx, nhwc16(x), w, b, y, residual=resid, rowvec=rowv, rv_bs=cout, rv_ss=B * cout, step=step)
    d.gn_groups = 32
    d.gno_gamma, d.gno_beta, d.gno_eps, d.gno_silu = gamma.data_ptr(), beta.data_ptr(), 1e-5, int(silu)
    d.gno_y, d.gno_ld, d.gno_skip_y = yn.data_ptr(), cout, int(skip_y)
    cpg = cout // 32
    try:
        ctx.conv_override(-1, sk)
        if cpg % 4 or H * W * cpg // 4 > 512:
            assert ctx.conv_gn_fused(d)[0] == 0
            ctx.conv(d)
            torch.cuda.synchronize()
            check(y.permute(0, 3, 1, 2), ref_h)
            assert not yn.any()
            return
        assert ctx.conv_gn_fused(d)[0] == 3
        ctx.conv(d)
        torch.cuda.synchronize()
        check(yn.permute(0, 3, 1, 2), ref_n)
        if skip_y:
            assert not y.any()
        else:
            check(y.permute(0, 3, 1, 2), ref_h)
        first = yn.clone()
        ctx.conv(d)
        torch.cuda.synchronize()
        assert torch.equal(first, yn)  # fixed-order reduction: bitwise reproducible
        ctx.conv_override(-1, 1)
        assert ctx.conv_gn_fused(d)[0] == 0
    finally:
        ctx.conv_override(-1, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("C,heads,dh,nq,nkv", [(224, 8, 28, 1024, 87), (448, 8, 56, 256, 87), (224, 8, 28, 768, 87),
                                                (448, 7, 64, 192, 77), (224, 7, 32, 100, 33)])
def test_attention_with_query_projection_inside(ctx, C, heads, dh, nq, nkv):
    """upk_attention_qproj_f16 = LayerNorm -> to_q -> cross-attention (attention.py:170-196 behind :213) against the
    same three steps in plain PyTorch (q rounded to fp16 as the unfused path's GEMM output is)."""
    from upgpt_amd.engine import head_pad, qproj_pack
    B = 2
    dp = head_pad(dh)
    x = (rnd(B, nq, C) * 1.5 + 0.2).half()
    w = rnd(heads * dh, C, seed=1, scale=1 / math.sqrt(C))
    gamma, beta = 1 + 0.2 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    k = torch.zeros(B, nkv, heads * dp, device=DEV, dtype=torch.float16)
    k.view(B, nkv, heads, dp)[..., :dh] = rnd(B, nkv, heads, dh, seed=4).half()
    v = rnd(B, nkv, heads, dh, seed=5).half()
    vt_ld = (nkv + 31) // 32 * 32
    vt = torch.zeros(B, heads, dp, vt_ld, device=DEV, dtype=torch.float16)
    vt[:, :, :dh, :nkv] = v.permute(0, 2, 3, 1)
    scale = dh ** -0.5
    wq, wu, wb = qproj_pack(w, gamma, beta, heads, dh, dp, C, DEV)
    out = torch.zeros(B, nq, heads * dp, device=DEV, dtype=torch.float16)
    ctx._chk(ctx.lib.upk_attention_qproj_f16(ctx.h, x.data_ptr(), C, nq * C, C, C, 1e-5, wq.data_ptr(), wu.data_ptr(),
                                             wb.data_ptr(), k.data_ptr(), heads * dp, nkv * heads * dp, vt.data_ptr(),
                                             vt_ld, out.data_ptr(), heads * dp, nq * heads * dp, B, heads, nq, nkv, dp,
                                             scale, ctx._s()))
    torch.cuda.synchronize()
    q = F.linear(F.layer_norm(x.float(), (C,), gamma, beta, 1e-5), w).half().float().view(B, nq, heads, dh).transpose(1, 2)
    kf = k.view(B, nkv, heads, dp)[..., :dh].float().transpose(1, 2)
    ref = torch.softmax(q @ kf.transpose(-1, -2) * scale, -1) @ v.float().transpose(1, 2)
    got = out.view(B, nq, heads, dp)[..., :dh].transpose(1, 2)
    check(got, ref, tol=1e-2)
    assert not out.view(B, nq, heads, dp)[..., dh:].any()  # the padded head columns stay zero


@pytest.mark.gpu
@pytest.mark.parametrize("M,C,N,flags", [(8192, 224, 768, 0), (2048, 448, 512, 0), (512, 896, 1024, 0), (512, 896, 7168, L.F_GEGLU),
                                         (200, 224, 256, 0)])
def test_layernorm_rows_from_producer(ctx, M, C, N, flags):
    """include/upk.h ln_rows_*: a residual GEMM leaves the LayerNorm row sums of its output, the folded-LayerNorm
    Linear behind it takes them from there — on every tile configuration family (wave-specialised M x N split, K-split,
    classic) — and matches LayerNorm + Linear (+ GEGLU) in PyTorch; producers that cannot say so."""
    import ctypes as ct
    K0 = 256
    a0 = rnd(M, K0).half()
    w0 = rnd(C, K0, seed=1, scale=1 / math.sqrt(K0))
    b0 = rnd(C, seed=2, scale=0.1)
    res = (rnd(M, C, seed=3) * 1.5 + 0.3).half()
    t = torch.zeros(M, C, device=DEV, dtype=torch.float16)
    dp = make_desc(ctx, a0.view(1, M, 1, K0), w0.view(C, K0, 1, 1), b0, t.view(1, M, 1, C), residual=res.view(1, M, 1, C))
    rows = torch.zeros(8, M, 2, device=DEV)
    dp.ln_rows_out = rows.data_ptr()
    gamma, beta = 1 + 0.2 * rnd(C, seed=4), 0.1 * rnd(C, seed=5)
    w1 = rnd(N, C, seed=6, scale=1 / math.sqrt(C))
    b1 = rnd(N, seed=7, scale=0.1)
    t_ref = (F.linear(a0.float(), w0.half().float(), b0) + res.float()).half().float()
    y_ref = F.linear(F.layer_norm(t_ref, (C,), gamma, beta, 1e-5), w1, b1)
    n_cols = N
    if flags & L.F_GEGLU:
        v, g = y_ref.chunk(2, dim=-1)
        y_ref, n_cols = v * F.gelu(g), N // 2
    wf = (w1 * gamma[None, :])
    bf = b1 + w1 @ beta
    row_map = None
    if flags & L.F_GEGLU:
        from upgpt_amd.engine import geglu_rows_map
        row_map = geglu_rows_map(N // 2).to(DEV).int()
    names = [ctx.lib.upk_conv_config_name(i).decode() for i in range(ctx.lib.upk_conv_num_configs())]
    fam = {"mxn": [i for i, n in enumerate(names) if "w" in n and "x1x1k" not in n],
           "ksplit": [i for i, n in enumerate(names) if "x1x1k" in n],
           "classic": [i for i, n in enumerate(names) if "w" not in n]}
    try:
        # producer on an M x N-split configuration: statistics available
        ran_p = None
        for cfg in fam["mxn"]:
            ctx.conv_override(cfg, 1)
            slots = ct.c_int()
            try:
                ctx._chk(ctx.lib.upk_conv_ln_rows(ctx.h, ct.byref(dp), ct.byref(slots)))
            except L.UpkError:
                continue
            if slots.value > 0:
                ctx.conv(dp)
                ran_p = slots.value
                break
        assert ran_p, "no M x N-split configuration took the producer"
        torch.cuda.synchronize()
        check(t, t_ref)
        tt = t.float()
        got = rows[:ran_p].sum(0)
        assert torch.allclose(got[:, 0], tt.sum(1), rtol=1e-3, atol=1e-2)
        assert torch.allclose(got[:, 1], (tt * tt).sum(1), rtol=1e-3, atol=1e-2)
        # a K-split producer leaves them too (one slot per N tile)
        for cfg in fam["ksplit"]:
            ctx.conv_override(cfg, 1)
            slots = ct.c_int(0)
            try:
                ctx._chk(ctx.lib.upk_conv_ln_rows(ctx.h, ct.byref(dp), ct.byref(slots)))
            except L.UpkError:
                continue
            if slots.value > 0:
                rows.zero_()
                t.zero_()
                ctx.conv(dp)
                torch.cuda.synchronize()
                check(t, t_ref)
                got = rows[:slots.value].sum(0)
                assert torch.allclose(got[:, 0], tt.sum(1), rtol=1e-3, atol=1e-2)
                assert torch.allclose(got[:, 1], (tt * tt).sum(1), rtol=1e-3, atol=1e-2)
                ran_p = slots.value
                break
        # a launch that splits K across workgroups cannot
        ctx.conv_override(-1, 2)
        slots = ct.c_int(5)
        ctx._chk(ctx.lib.upk_conv_ln_rows(ctx.h, ct.byref(dp), ct.byref(slots)))
        assert slots.value == 0
        # consumer on every family
        for name, cfgs in fam.items():
            ran = 0
            for cfg in cfgs:
                y = torch.zeros(M, n_cols, device=DEV, dtype=torch.float16)
                dc = make_desc(ctx, t.view(1, M, 1, C), wf.view(N, C, 1, 1), bf, y.view(1, M, 1, n_cols), flags=flags,
                               n_out=n_cols if flags & L.F_GEGLU else None, row_map=row_map)
                colsum = torch.zeros(dc.n_pad, device=DEV)
                wq = wf.half().float().sum(1)
                if row_map is not None:
                    m = row_map.long()
                    colsum[: m.numel()] = torch.where(m >= 0, wq[m.clamp(min=0)], torch.zeros_like(wq[m.clamp(min=0)]))
                else:
                    colsum[:N] = wq
                dc.ln_colsum, dc.ln_eps, dc.ln_dim = colsum.data_ptr(), 1e-5, C
                dc.ln_rows_in, dc.ln_rows_slots = rows.data_ptr(), ran_p
                ctx.conv_override(cfg, 1)
                try:
                    ctx.conv(dc)
                except L.UpkError:
                    continue
                torch.cuda.synchronize()
                check(y, y_ref, tol=3e-2)
                ran += 1
                if ran >= 3:
                    break
            assert ran >= 1 or name != "mxn", "no %s configuration ran the consumer" % name
    finally:
        ctx.conv_override(-1, 0)


def _phase_weights(ctx, w):
    """The four 2x2 phase weights of nearest-2x -> conv3x3 (include/upk.h w_phase), from the definition: 3x3 tap k of
    output parity p reads low-resolution offset (p + k - 1) >> 1, phase tap t reads offset p + t - 1."""
    parts = []
    n_pad = None
    for py in (0, 1):
        for px in (0, 1):
            wp = torch.zeros(w.shape[0], w.shape[1], 2, 2, device=DEV)
            for ky in range(3):
                for kx in range(3):
                    ty, tx = ((py + ky - 1) >> 1) - (py - 1), ((px + kx - 1) >> 1) - (px - 1)
                    wp[:, :, ty, tx] += w[:, :, ky, kx]
            packed, n_pad = ctx.pack_weight(wp.contiguous())
            parts.append(packed.reshape(-1))
    return torch.cat(parts).contiguous(), n_pad


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,N,H,W,sk,f32", [(2, 64, 96, 8, 8, 1, False), (2, 224, 224, 16, 12, 1, False),
                                              (8, 896, 896, 4, 4, 4, False), (1, 128, 64, 33, 17, 1, False),
                                              (2, 96, 32, 8, 8, 2, True)])
def test_upsample_conv_as_four_phase_convs(ctx, B, C, N, H, W, sk, f32):
    """Upsample (openaimodel.py:109-119): F.interpolate(2x nearest) -> conv3x3, computed as four 2x2 convs on the
    low-resolution grid (upk_conv_desc.w_phase) — against the direct form in PyTorch and against the library's own
    full-resolution launch."""
    x = rnd(B, C, H, W)
    w = rnd(N, C, 3, 3, scale=1 / math.sqrt(9 * C))
    b = rnd(N, scale=0.1)
    ref = F.conv2d(F.interpolate(x.half().float(), scale_factor=2, mode="nearest"), w.half().float(), b, padding=1)
    xn = nhwc16(x)
    wph, n_pad = _phase_weights(ctx, w)
    outs = []
    for phased in (True, False):
        y = torch.zeros(B, 2 * H, 2 * W, N, device=DEV, dtype=torch.float32 if f32 else torch.float16)
        d = make_desc(ctx, xn, w, b, y, flags=L.F_UPSAMPLE2X | (L.F_OUT_F32 if f32 else 0))
        assert d.n_pad == n_pad
        if phased:
            d.w_phase = wph.data_ptr()
        try:
            ctx.conv_override(-1, sk)
            ctx.conv(d)
        finally:
            ctx.conv_override(-1, 0)
        torch.cuda.synchronize()
        check(y.permute(0, 3, 1, 2), ref)
        outs.append(y)
    # the two forms differ only by the fp16 rounding of the summed taps
    check(outs[0], outs[1], tol=1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,cin,cout,silu", [(64, 64, 64, 128, True), (96, 64, 32, 256, False), (128, 128, 32, 128, True)])
def test_groupnorm_from_many_block_partials(ctx, H, W, cin, cout, silu):
    """VAE-decoder-sized feature maps: the conv leaves per-(row block, channel) partials for more than 32 blocks per
    sample (gn_stats_cap), upk_groupnorm_finalize_f32 folds them, the apply pass normalises without a statistics
    pass (model.py:38-39, 82-121)."""
    B = 2
    x = rnd(B, cin, H, W)
    w = rnd(cout, cin, 3, 3, scale=1 / math.sqrt(9 * cin))
    b = rnd(cout, scale=0.1)
    gamma, beta = rnd(cout, seed=3) * 0.5 + 1.0, rnd(cout, seed=4) * 0.3
    y = torch.zeros(B, H, W, cout, device=DEV, dtype=torch.float16)
    d = make_desc(ctx, nhwc16(x), w, b, y)
    cap = H * W // 64
    sws = torch.zeros(ctx.gn_stats_floats(B, d.n_pad, cap), device=DEV)
    d.gn_stats_ws, d.gn_groups, d.gn_stats_cap = sws.data_ptr(), 32, cap
    mode, nblk = ctx.conv_gn_fused(d)
    assert mode == 2 and 32 < nblk <= cap, (mode, nblk)
    ctx.conv(d)
    ws = torch.zeros(ctx.groupnorm_ws_bytes(B, H * W) // 4 + 64, device=DEV)
    ctx._chk(ctx.lib.upk_groupnorm_finalize_f32(ctx.h, sws.data_ptr(), nblk, d.n_pad, B, H * W, cout, 32, ws.data_ptr(),
                                                ctx._s()))
    yn = torch.zeros_like(y)
    ctx._chk(ctx.lib.upk_groupnorm_apply_nhwc_f16(ctx.h, y.data_ptr(), cout, cout, None, 0, 0, B, H * W, 32,
                                                  gamma.data_ptr(), beta.data_ptr(), 1e-6, int(silu), yn.data_ptr(), cout,
                                                  ws.data_ptr(), 1, 0, 0, None, 0, 0, ctx._s()))
    torch.cuda.synchronize()
    ref = F.group_norm(y.float().permute(0, 3, 1, 2), 32, gamma, beta, 1e-6)
    if silu:
        ref = F.silu(ref)
    check(yn.permute(0, 3, 1, 2), ref)
    # without the capacity field the launch reports no by-product (33+ blocks do not fit the default buffer)
    d.gn_stats_cap = 0
    assert ctx.conv_gn_fused(d)[0] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,cin,cout,ks", [(8, 8, 8, 128, 256, 3), (8, 16, 16, 64, 448, 1), (4, 4, 4, 256, 896, 3)])
def test_conv_xcd_aware_tile_order_visits_every_tile(ctx, B, H, W, cin, cout, ks):
    """Grids large enough for the XCD-aware tile order (IgemmArgs::xm_*: padded grid, (M tiles) x (N tile, K split) units
    dealt to a pm x pn arrangement of the 8 XCDs): every configuration x split-K must still write every output tile
    exactly as F.conv2d does (a skipped or doubled tile shows as zeros / a wrong split-K sum)."""
    x = rnd(B, cin, H, W)
    w = rnd(cout, cin, ks, ks, scale=1 / math.sqrt(ks * ks * cin))
    b = rnd(cout, scale=0.1)
    ref = F.conv2d(x.half().float(), w.half().float(), b, padding=ks // 2)
    xn = nhwc16(x)
    ran = 0
    try:
        for cfg in range(ctx.lib.upk_conv_num_configs()):
            for sk in (1, 2, 4, 8):
                y = torch.full((B, H, W, cout), float("nan"), device=DEV, dtype=torch.float16)
                ctx.conv_override(cfg, sk)
                try:
                    ctx.conv(make_desc(ctx, xn, w, b, y))
                except L.UpkError:
                    continue
                torch.cuda.synchronize()
                assert torch.isfinite(y).all(), (cfg, sk)
                check(y.permute(0, 3, 1, 2), ref)
                ran += 1
    finally:
        ctx.conv_override(-1, 0)
    assert ran >= 60

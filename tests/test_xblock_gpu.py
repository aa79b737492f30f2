"""Fused cross-attention half of a BasicTransformerBlock (upgpt_amd/csrc/xblock.hip, include/upk.h upk_cross_block_f16)
through the C ABI against a plain PyTorch fp32 reference of attn1.to_out (+ x) -> norm2 -> attn2 over the context
(+ x) (attention.py:178-192, 257-260), and the engine's use of it inside the UNet forward against the unfused path."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from upgpt_amd import _lib as L
from test_ops_gpu import DEV, check, rnd

pytestmark = pytest.mark.gpu


def head_cols(heads, dh, dp):
    """Column map of the padded head layout: packed column h * dp + j <- real column h * dh + j (j < dh), else -1."""
    m = torch.full((heads * dp,), -1, dtype=torch.int32)
    for h in range(heads):
        m[h * dp: h * dp + dh] = torch.arange(h * dh, (h + 1) * dh, dtype=torch.int32)
    return m


@pytest.mark.parametrize("B,hw,c,dh,dp,rows,nkv", [(8, 1024, 224, 28, 32, 32, 87), (2, 64, 224, 28, 32, 16, 87),
                                                   (3, 256, 448, 56, 64, 32, 77), (8, 256, 448, 56, 64, 16, 96),
                                                   (1, 32, 224, 28, 32, 32, 1)])
def test_cross_block_vs_torch(ctx, B, hw, c, dh, dp, rows, nkv):
    heads = 8
    M, hd, inner = B * hw, heads * dp, heads * dh
    a1r = rnd(M, inner, seed=1)                      # self-attention output, real head width
    t0 = (rnd(M, c, seed=2) * 1.5 + 0.3).half()
    wo1, bo1 = rnd(c, inner, scale=1 / math.sqrt(inner), seed=3), rnd(c, scale=0.1, seed=4)
    gamma, beta = 1 + 0.2 * rnd(c, seed=5), 0.1 * rnd(c, seed=6)
    wq = rnd(inner, c, scale=1 / math.sqrt(c), seed=7)
    wo2, bo2 = rnd(c, inner, scale=1 / math.sqrt(inner), seed=8), rnd(c, scale=0.1, seed=9)
    kr, vr = rnd(B, nkv, inner, seed=10), rnd(B, nkv, inner, seed=11)
    scale = dh ** -0.5
    cols = head_cols(heads, dh, dp).to(DEV)
    real = cols >= 0

    def padded(x):  # [..., inner] -> [..., hd] fp16
        out = torch.zeros(*x.shape[:-1], hd, device=DEV, dtype=torch.float16)
        out[..., real] = x.half()
        return out

    a1 = padded(a1r)
    kc = padded(kr).reshape(B * nkv, hd).contiguous()
    vt_ld = 96
    vt = torch.zeros(B, heads, dp, vt_ld, device=DEV, dtype=torch.float16)
    vt[:, :, :dh, :nkv] = vr.half().reshape(B, nkv, heads, dh).permute(0, 2, 3, 1)
    # reference (fp32 on the fp16-rounded operands; t1 is rounded to fp16 as the kernel keeps it)
    t1 = (a1r.half().float() @ wo1.half().float().t() + bo1 + t0.float()).half().float()
    wqf = (wq * gamma[None, :]).half().float()
    xn = (t1 - t1.mean(1, keepdim=True)) * torch.rsqrt(t1.var(1, unbiased=False, keepdim=True) + 1e-5)
    q = (xn @ wqf.t() + wq @ beta).half().float().reshape(B, hw, heads, dh).permute(0, 2, 1, 3)
    k = kr.half().float().reshape(B, nkv, heads, dh).permute(0, 2, 1, 3)
    v = vr.half().float().reshape(B, nkv, heads, dh).permute(0, 2, 1, 3)
    p = torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1)
    a2 = (p @ v).permute(0, 2, 1, 3).reshape(M, inner).half().float()
    ref = a2 @ wo2.half().float().t() + bo2 + t1
    # operands
    w1p, n1 = ctx.pack_weight(wo1.contiguous(), col_map=cols)
    w3p, n3 = ctx.pack_weight(wo2.contiguous(), col_map=cols)
    wqp, nq = ctx.pack_weight((wq * gamma[None, :]).contiguous(), row_map=cols)
    assert n1 == c and n3 == c and nq == hd
    uq = torch.zeros(hd, device=DEV); uq[real] = wqf.sum(dim=1)
    bq = torch.zeros(hd, device=DEV); bq[real] = wq @ beta
    vec = torch.cat([bo1, uq, bq, bo2])
    vec = torch.cat([vec, vec.new_zeros(-vec.numel() % 256)]).contiguous()
    y = torch.zeros(M, c, device=DEV, dtype=torch.float16)
    d = L.XblockDesc()
    d.a1, d.lda, d.m, d.c, d.heads, d.d = a1.data_ptr(), hd, M, c, heads, dp
    d.t0, d.ld_t0 = t0.data_ptr(), c
    d.w_out1, d.w_q, d.w_out2, d.vec = w1p.data_ptr(), wqp.data_ptr(), w3p.data_ptr(), vec.data_ptr()
    d.ln_eps, d.ln_dim = 1e-5, c
    d.k_ctx, d.ldk, d.n_kv = kc.data_ptr(), hd, nkv
    d.vt_ctx, d.vt_ld, d.scale = vt.data_ptr(), vt_ld, scale
    d.y, d.ldy, d.hw, d.rows_per_wg = y.data_ptr(), c, hw, rows
    assert ctx.lib.upk_cross_block_supported(ctx.h, C.byref(d))
    ctx._chk(ctx.lib.upk_cross_block_f16(ctx.h, C.byref(d), ctx._s()))
    torch.cuda.synchronize()
    check(y, ref, tol=8e-3)
    y2 = torch.zeros_like(y)
    d.y = y2.data_ptr()
    ctx._chk(ctx.lib.upk_cross_block_f16(ctx.h, C.byref(d), ctx._s()))
    torch.cuda.synchronize()
    assert torch.equal(y, y2)


def test_cross_block_refuses_shapes_outside_its_domain(ctx):
    d = L.XblockDesc()
    d.m, d.c, d.heads, d.d, d.hw, d.rows_per_wg, d.n_kv, d.vt_ld = 1024, 896, 8, 128, 64, 32, 87, 96
    d.lda, d.ld_t0, d.ldy, d.ldk = 1024, 896, 896, 1024
    assert not ctx.lib.upk_cross_block_supported(ctx.h, C.byref(d))
    d.c, d.d, d.lda, d.ldk, d.ld_t0, d.ldy, d.n_kv = 224, 32, 256, 256, 224, 224, 120
    assert not ctx.lib.upk_cross_block_supported(ctx.h, C.byref(d))  # more context keys than the kernel covers
    buf = torch.zeros(16, device=DEV)
    for f in ("a1", "t0", "w_out1", "w_q", "w_out2", "vec", "k_ctx", "vt_ctx", "y"):
        setattr(d, f, buf.data_ptr())
    with pytest.raises(L.UpkError):
        ctx._chk(ctx.lib.upk_cross_block_f16(ctx.h, C.byref(d), ctx._s()))


def test_unet_forward_with_and_without_the_fused_cross_block():
    """The bbox UNet at the bench shape (B = 8, 32x32): eps with the fused cross-attention half against the unfused
    launches (UPGPT_XBLOCK=0), same weights and inputs."""
    import upgpt_amd
    from upgpt_amd import knobs, synth

    def run(mode):
        old = knobs.XBLOCK
        knobs.XBLOCK = mode
        try:
            m = upgpt_amd.build_model("bbox")
            synth.fill_module_(m)
            m = m.cuda()
            inp = synth.synth_inputs(8, (32, 32), 4, 87, 768, seed=3, text_only=True)
            cond = {"c_crossattn": inp["c_crossattn"].cuda(), "c_concat": [inp["c_concat"].cuda()]}
            t = torch.full((8,), 601, dtype=torch.long, device=DEV)
            eps = m.apply_model(inp["x_T"].cuda(), t, cond)
            pl = next(iter(m.model.diffusion_model._plans.values()))
            return eps.float().cpu(), sum(1 for lab in pl.body.labels if lab.startswith("xblock "))
        finally:
            knobs.XBLOCK = old

    e1, n1 = run("1")
    e0, n0 = run("0")
    assert n0 == 0 and n1 >= 5, (n0, n1)
    assert float(((e1 - e0) ** 2).mean()) < 1e-5 * max(1.0, float((e0 ** 2).mean()))


@pytest.mark.parametrize("B,hw,rows", [(8, 1024, 32), (2, 64, 16), (1, 32, 32)])
def test_head_block_vs_torch(ctx, B, hw, rows):
    """upk_head_block_f16 (proj_in -> norm1 -> q | k | v, V transposed) against fp32 PyTorch."""
    heads, c, dh, dp = 8, 224, 28, 32
    M, hd, inner = B * hw, heads * dp, heads * dh
    x = (rnd(M, c, seed=1) * 1.3).half()
    wi, bi = rnd(c, c, scale=1 / math.sqrt(c), seed=2), rnd(c, scale=0.1, seed=3)
    gamma, beta = 1 + 0.2 * rnd(c, seed=4), 0.1 * rnd(c, seed=5)
    wqkv = rnd(3 * inner, c, scale=1 / math.sqrt(c), seed=6)
    cols = head_cols(heads, dh, dp).to(DEV)
    rows3 = torch.cat([torch.where(cols >= 0, cols + i * inner, cols) for i in range(3)]).to(torch.int32)
    real3 = rows3 >= 0
    # reference
    t0 = (x.float() @ wi.half().float().t() + bi).half().float()
    wf = (wqkv * gamma[None, :]).half().float()
    xn = (t0 - t0.mean(1, keepdim=True)) * torch.rsqrt(t0.var(1, unbiased=False, keepdim=True) + 1e-5)
    qkv = xn @ wf.t() + wqkv @ beta
    # operands
    w1p, n1 = ctx.pack_weight(wi.contiguous())
    w2p, n2 = ctx.pack_weight((wqkv * gamma[None, :]).contiguous(), row_map=rows3)
    assert n1 == c and n2 == 3 * hd
    u2 = torch.zeros(3 * hd, device=DEV); u2[real3] = wf.sum(dim=1)
    b2 = torch.zeros(3 * hd, device=DEV); b2[real3] = wqkv @ beta
    vec = torch.cat([bi, u2, b2])
    vec = torch.cat([vec, vec.new_zeros(-vec.numel() % 256)]).contiguous()
    vt_ld = (hw + 31) // 32 * 32
    t0o = torch.zeros(M, c, device=DEV, dtype=torch.float16)
    qk = torch.zeros(M, 2 * hd, device=DEV, dtype=torch.float16)
    vt = torch.zeros(B, heads, dp, vt_ld, device=DEV, dtype=torch.float16)
    d = L.HblockDesc()
    d.x, d.ldx, d.m, d.c, d.heads, d.d = x.data_ptr(), c, M, c, heads, dp
    d.w_in, d.w_qkv, d.vec, d.ln_eps, d.ln_dim = w1p.data_ptr(), w2p.data_ptr(), vec.data_ptr(), 1e-5, c
    d.t0, d.ld_t0, d.qk, d.ld_qk, d.vt, d.vt_ld = t0o.data_ptr(), c, qk.data_ptr(), 2 * hd, vt.data_ptr(), vt_ld
    d.hw, d.rows_per_wg = hw, rows
    assert ctx.lib.upk_head_block_supported(ctx.h, C.byref(d))
    ctx._chk(ctx.lib.upk_head_block_f16(ctx.h, C.byref(d), ctx._s()))
    torch.cuda.synchronize()
    check(t0o, t0, tol=4e-3)
    ref_qk = torch.zeros(M, 2 * hd, device=DEV)
    ref_qk[:, real3[: 2 * hd]] = qkv[:, : 2 * inner]
    check(qk, ref_qk, tol=8e-3)
    ref_v = torch.zeros(M, hd, device=DEV)
    ref_v[:, cols >= 0] = qkv[:, 2 * inner:]
    ref_vt = ref_v.reshape(B, hw, heads, dp).permute(0, 2, 3, 1)
    check(vt[..., :hw], ref_vt, tol=8e-3)
    assert float(vt[..., hw:].abs().max()) == 0.0 if vt_ld > hw else True


@pytest.mark.parametrize("B,hw,rows,nblk", [(8, 1024, 32, 16), (2, 64, 16, 4)])
def test_head_block_groupnorm_fold_is_bit_identical_to_the_launch(ctx, B, hw, rows, nblk):
    """gn_part: the GroupNorm of the input applied on the tile inside the kernel == upk_groupnorm_apply + plain call."""
    heads, c, dp = 8, 224, 32
    M, hd = B * hw, heads * dp
    x = (rnd(M, c, seed=1) * 1.7 + 0.4).half()
    gamma, beta = 1 + 0.3 * rnd(c, seed=2), 0.2 * rnd(c, seed=3)
    w1p, _ = ctx.pack_weight(rnd(c, c, scale=1 / math.sqrt(c), seed=4).contiguous())
    w2p, _ = ctx.pack_weight(rnd(3 * hd, c, scale=1 / math.sqrt(c), seed=5).contiguous())
    vec = torch.cat([rnd(c, scale=0.1, seed=6), rnd(3 * hd, scale=0.3, seed=7), rnd(3 * hd, scale=0.1, seed=8)])
    vec = torch.cat([vec, vec.new_zeros(-vec.numel() % 256)]).contiguous()
    # per-(row block, channel) partials as a producer's epilogue leaves them: [B][nblk][2][ld]
    ld = c
    xf = x.float().reshape(B, nblk, hw // nblk, c)
    part = torch.stack([xf.sum(2), (xf * xf).sum(2)], dim=2).contiguous()
    vt_ld = (hw + 31) // 32 * 32
    outs = []
    for fold in (False, True):
        t0 = torch.zeros(M, c, device=DEV, dtype=torch.float16)
        qk = torch.zeros(M, 2 * hd, device=DEV, dtype=torch.float16)
        vt = torch.zeros(B, heads, dp, vt_ld, device=DEV, dtype=torch.float16)
        d = L.HblockDesc()
        d.m, d.c, d.heads, d.d = M, c, heads, dp
        d.w_in, d.w_qkv, d.vec, d.ln_eps, d.ln_dim = w1p.data_ptr(), w2p.data_ptr(), vec.data_ptr(), 1e-5, c
        d.t0, d.ld_t0, d.qk, d.ld_qk, d.vt, d.vt_ld = t0.data_ptr(), c, qk.data_ptr(), 2 * hd, vt.data_ptr(), vt_ld
        d.hw, d.rows_per_wg = hw, rows
        if fold:
            d.x, d.ldx = x.data_ptr(), c
            d.gn_part, d.gn_gamma, d.gn_beta = part.data_ptr(), gamma.data_ptr(), beta.data_ptr()
            d.gn_nblk, d.gn_ld, d.gn_groups, d.gn_eps = nblk, ld, 32, 1e-6
        else:
            xn = torch.zeros_like(x)
            ctx._chk(ctx.lib.upk_groupnorm_apply_nhwc_f16(ctx.h, x.data_ptr(), c, c, None, 0, 0, B, hw, 32, gamma.data_ptr(),
                                                          beta.data_ptr(), 1e-6, 0, xn.data_ptr(), c, part.data_ptr(), 2, nblk,
                                                          ld, None, 0, 0, ctx._s()))
            d.x, d.ldx = xn.data_ptr(), c
        ctx._chk(ctx.lib.upk_head_block_f16(ctx.h, C.byref(d), ctx._s()))
        torch.cuda.synchronize()
        outs.append((t0, qk, vt))
    for u, v in zip(*outs):
        assert torch.equal(u, v)
    ref = F.group_norm(x.float().reshape(B, hw, c).permute(0, 2, 1), 32, gamma, beta, 1e-6).permute(0, 2, 1).reshape(M, c)
    assert float(ref.abs().max()) > 1.0  # (sanity of the test data)


def test_unet_forward_with_and_without_the_fused_head():
    import upgpt_amd
    from upgpt_amd import knobs, synth

    def run(mode):
        old = knobs.HBLOCK
        knobs.HBLOCK = mode
        try:
            m = upgpt_amd.build_model("bbox")
            synth.fill_module_(m)
            m = m.cuda()
            inp = synth.synth_inputs(8, (32, 32), 4, 87, 768, seed=3, text_only=True)
            cond = {"c_crossattn": inp["c_crossattn"].cuda(), "c_concat": [inp["c_concat"].cuda()]}
            t = torch.full((8,), 601, dtype=torch.long, device=DEV)
            eps = m.apply_model(inp["x_T"].cuda(), t, cond)
            pl = next(iter(m.model.diffusion_model._plans.values()))
            return eps.float().cpu(), sum(1 for lab in pl.body.labels if lab.startswith("hblock "))
        finally:
            knobs.HBLOCK = old

    e1, n1 = run("auto")
    e0, n0 = run("0")
    assert n0 == 0 and n1 == 5, (n0, n1)
    assert float(((e1 - e0) ** 2).mean()) < 1e-5 * max(1.0, float((e0 ** 2).mean()))


def test_head_block_behind_a_split_k_producer():
    """which GroupNorm path a fused SpatialTransformer head takes is decided by the TUNED configuration of
    the conv in front of it (upk_conv_gn_fused: 2 = channel partials of an unsplit launch -> normalisation inside the head;
    1 = group partials of a split-K reduce -> apply launch + plain head; 3 = the reduce pass normalises itself, reachable
    at the 32x32 level only with the dev knob UPK_GNAPPLY_NVMAX raised).  A tuning change could flip the path silently, so
    this test forces the other one: every producer of a head's input is pinned to split-K = 2, and the forward must agree
    with the separate-GroupNorm reference path (HBLOCK_GN off)."""
    import ctypes as C
    import upgpt_amd
    from upgpt_amd import knobs, synth
    m = upgpt_amd.build_model("bbox")
    synth.fill_module_(m)
    m = m.cuda()
    inp = synth.synth_inputs(8, (32, 32), 4, 87, 768, seed=3, text_only=True)
    cond = {"c_crossattn": inp["c_crossattn"].cuda(), "c_concat": [inp["c_concat"].cuda()]}
    t = torch.full((8,), 601, dtype=torch.long, device=DEV)
    unet = m.model.diffusion_model

    def run(gn_fold, force_split):
        old = knobs.HBLOCK_GN
        knobs.HBLOCK_GN = gn_fold
        try:
            for pl in unet._plans.values():
                pl.close()
            unet._plans.clear()
            pl = unet.plan(8, 32, 32, 87, 8, "forward")
            heads = [i for i, lab in enumerate(pl.body.labels) if lab.startswith("hblock ")]
            assert len(heads) == 5
            forced = 0
            if force_split:
                for i in heads:
                    j = max(k for k in range(i) if pl.body.meta[k] is not None)  # the conv that wrote the head's input
                    d = pl.body.meta[j]
                    d.tune_splitk = 2
                    mode, nblk = C.c_int(0), C.c_int(0)
                    pl.ctx._chk(pl.lib.upk_conv_gn_fused(pl.hctx, C.byref(d), C.byref(mode), C.byref(nblk)))
                    forced += mode.value in (1, 3)
            eps = m.apply_model(inp["x_T"].cuda(), t, cond)
            return eps.float().cpu(), forced
        finally:
            knobs.HBLOCK_GN = old

    e_ref, _ = run(False, False)
    e_m3, forced = run(True, True)
    assert forced == 5, "the forced split-K did not move the heads' producers off the channel-partials path"
    assert float(((e_m3 - e_ref) ** 2).mean()) < 1e-5 * max(1.0, float((e_ref ** 2).mean()))

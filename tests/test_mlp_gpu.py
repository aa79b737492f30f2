"""Fused feed-forward tail (upgpt_amd/csrc/mlp.hip, include/upk.h upk_geglu_mlp_f16) through the C ABI against a plain
PyTorch fp32 reference of norm3 -> GEGLU -> ff.net.2 (+ residual) -> proj_out (+ x_in) (attention.py:42-64, 215, 259-261),
and the engine's use of it inside the UNet forward against the two-launch path."""
import ctypes as C
import math
import os

import pytest
import torch
import torch.nn.functional as F

from upgpt_amd import _lib as L
from test_ops_gpu import DEV, check, geglu_row_map, rnd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,c,rows,hw", [(8192, 224, 64, 1024), (8192, 224, 32, 1024), (1000, 224, 64, 0), (2048, 448, 32, 256),
                                         (96, 224, 32, 0)])
def test_geglu_mlp_vs_torch(ctx, M, c, rows, hw):
    inner = 4 * c
    g = torch.Generator(device="cpu").manual_seed(5)
    x = (torch.randn(M, c, generator=g) * (0.5 + 2 * torch.rand(M, 1, generator=g)) + 2 * torch.randn(M, 1, generator=g)).to(DEV).half()
    res = rnd(M, c, seed=7).half()
    gamma, beta = 1 + 0.2 * rnd(c, seed=2), 0.1 * rnd(c, seed=3)
    w1 = rnd(2 * inner, c, scale=1 / math.sqrt(c), seed=4)
    b1 = rnd(2 * inner, scale=0.1, seed=5)
    w2h = rnd(c, inner, scale=1 / math.sqrt(inner), seed=6)   # (P F2)
    w2x = rnd(c, c, scale=1 / math.sqrt(c), seed=8)           # P
    b2 = rnd(c, scale=0.1, seed=9)
    # reference
    hpre = F.layer_norm(x.float(), (c,), gamma, beta, 1e-5) @ w1.half().float().t() * 0  # (shape only)
    w1f = (w1 * gamma[None, :]).half().float()
    xn = (x.float() - x.float().mean(1, keepdim=True)) * torch.rsqrt(x.float().var(1, unbiased=False, keepdim=True) + 1e-5)
    hpre = xn @ w1f.t() + (b1 + w1 @ beta)
    h = (hpre[:, :inner] * F.gelu(hpre[:, inner:])).half().float()
    ref = res.float() + h @ w2h.half().float().t() + x.float() @ w2x.half().float().t() + b2
    # operands
    rm = geglu_row_map(inner).to(DEV)
    w1p, n1 = ctx.pack_weight((w1 * gamma[None, :]).contiguous(), row_map=rm)
    b1p = (b1 + w1 @ beta)[rm.long()].contiguous()
    u1p = (w1 * gamma[None, :]).half().float().sum(dim=1)[rm.long()].contiguous()
    w2a, n_pad = ctx.pack_weight(w2h.contiguous())
    w2b, _ = ctx.pack_weight(w2x.contiguous())
    w2p = torch.cat([w2a.reshape(-1), w2b.reshape(-1)]).contiguous()
    b2p = torch.zeros(n_pad, device=DEV); b2p[:c] = b2
    y = torch.zeros(M, c, device=DEV, dtype=torch.float16)
    d = L.MlpDesc()
    d.x, d.ldx, d.m, d.c, d.inner = x.data_ptr(), c, M, c, inner
    d.w1, d.b1, d.u1, d.ln_eps, d.ln_dim = w1p.data_ptr(), b1p.data_ptr(), u1p.data_ptr(), 1e-5, c
    d.w2, d.b2, d.n_out, d.n_pad = w2p.data_ptr(), b2p.data_ptr(), c, n_pad
    d.residual, d.ld_res, d.y, d.ldy = res.data_ptr(), c, y.data_ptr(), c
    d.rows_per_wg, d.hw = rows, hw
    sws = None
    if hw:
        B = M // hw
        sws = torch.zeros(ctx.gn_stats_floats(B, n_pad), device=DEV)
        d.gn_stats_ws = sws.data_ptr()
    if not ctx.lib.upk_geglu_mlp_supported(ctx.h, C.byref(d)):
        assert c != 224 or n_pad > 256, "the 224-channel shapes must be supported"
        with pytest.raises(L.UpkError):
            ctx._chk(ctx.lib.upk_geglu_mlp_f16(ctx.h, C.byref(d), ctx._s()))
        return
    ctx._chk(ctx.lib.upk_geglu_mlp_f16(ctx.h, C.byref(d), ctx._s()))
    torch.cuda.synchronize()
    check(y, ref, tol=8e-3)
    if hw:
        nblk = hw // rows
        part = sws[: B * nblk * 2 * n_pad].reshape(B, nblk, 2, n_pad)
        yf = y.float().reshape(B, hw, c)
        check(part[:, :, 0, :c].sum(1), yf.sum(1), tol=2e-3)
        check(part[:, :, 1, :c].sum(1), (yf * yf).sum(1), tol=2e-3)
    # bitwise reproducible
    y2 = torch.zeros_like(y)
    d.y = y2.data_ptr()
    ctx._chk(ctx.lib.upk_geglu_mlp_f16(ctx.h, C.byref(d), ctx._s()))
    torch.cuda.synchronize()
    assert torch.equal(y, y2)


def test_unet_forward_with_and_without_the_fused_mlp():
    """The bbox UNet at the bench shape (B = 8, 32x32): eps with the fused feed-forward tail (default where it covers the
    chip) against the two-launch path (UPGPT_MLP_FUSE=0), same weights and inputs."""
    import importlib
    import upgpt_amd
    from upgpt_amd import knobs, synth

    def run(mode):
        old = knobs.MLP_FUSE
        knobs.MLP_FUSE = mode
        try:
            m = upgpt_amd.build_model("bbox")
            synth.fill_module_(m)
            m = m.cuda()
            inp = synth.synth_inputs(8, (32, 32), 4, 87, 768, seed=3, text_only=True)
            cond = {"c_crossattn": inp["c_crossattn"].cuda(), "c_concat": [inp["c_concat"].cuda()]}
            t = torch.full((8,), 601, dtype=torch.long, device=DEV)
            eps = m.apply_model(inp["x_T"].cuda(), t, cond)
            pl = next(iter(m.model.diffusion_model._plans.values()))
            return eps.float().cpu(), sum(1 for lab in pl.body.labels if lab.startswith("mlp "))
        finally:
            knobs.MLP_FUSE = old

    e1, n1 = run("auto")
    e0, n0 = run("0")
    assert n0 == 0 and n1 == 5, (n0, n1)  # the five 32x32-level transformer blocks
    assert float(((e1 - e0) ** 2).mean()) < 1e-5 * max(1.0, float((e0 ** 2).mean()))

"""Per-XCD persistent engine (upgpt_amd/csrc/xcd.hip, include/upk.h upk_xcd_run_f16): a whole SpatialTransformer
(attention.py:250-261) as one launch, sample b on XCD b % 8, XCD-local barriers between its ten phases.

Parity: the UNet forward with every transformer on the engine against (1) the golden outputs of the REAL reference,
(2) the CPU oracle, (3) the launch-chain path of the same library block by block (fp16 rounding differs between the two
paths — norm affines folded into weights, LayerNorm applied to the staged rows — so (3) is a tolerance, (1) / (2) are the
parity bars of tests/test_model_gpu.py).  Plus the protocol's own properties: the synchronisation words come back zeroed,
status stays 0, reruns are bit-identical, idle XCDs (B < 8) and several samples per XCD (B > 8) are covered."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import upgpt_amd
from oracle import unet as o_unet
from upgpt_amd import engine, knobs, synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
KIND = {"tiny": dict(unet=synth.TINY_UNET, C=4, ntok=87, cc=1), "bbox": dict(unet=synth.BBOX_UNET, C=4, ntok=87, cc=1),
        "upscale": dict(unet=synth.UPSCALE_UNET, C=3, ntok=86, cc=3)}
_cache = {}


def get_model(kind, monkeypatch):
    monkeypatch.setattr(knobs, "XCD", "1")  # (the engine's operands are packed with the weights)
    if kind not in _cache:
        m = upgpt_amd.build_model(kind, overrides={"image_size": [32, 24]} if kind == "upscale" else None)
        sd = synth.fill_module_(m)
        _cache[kind] = (m.cuda(), sd)
    return _cache[kind]


def forward(model, mode, x, t, ctx, monkeypatch, want_taps=False):
    """UNet forward with the engine forced on ("1") / off ("0"); returns (eps, taps, #xcd ops, plan)."""
    unet = model.model.diffusion_model
    monkeypatch.setattr(knobs, "XCD", mode)
    for pl in unet._plans.values():
        pl.close()
    unet._plans.clear()
    eps = unet(x.cuda(), t.cuda(), context=ctx.cuda())
    pl = next(iter(unet._plans.values()))
    taps = {k: v.t.float().cpu() for k, v in pl.taps.items()} if want_taps else None
    n_x = sum(1 for lab in pl.body.labels if lab.startswith("xcd "))
    return eps, taps, n_x, pl


def xcd_status(pl):
    st = C.c_int(-1)
    pl.ctx._chk(pl.lib.upk_xcd_status(pl.hctx, pl.xcd_sync.data_ptr(), C.byref(st)))
    return st.value


def mse(a, b):
    return float(((a.float().cpu() - torch.as_tensor(b).float().cpu()) ** 2).mean())


@pytest.mark.parametrize("B", [3, 9])
def test_tiny_unet_on_the_engine_vs_chain_and_oracle(B, monkeypatch):
    """B = 3: XCDs 3-7 idle; B = 9: XCD 0 works on two samples."""
    model, sd = get_model("tiny", monkeypatch)
    inp = synth.synth_inputs(B, (32, 24), 4, 87, 768, seed=11)
    x = torch.cat([inp["x_T"], inp["c_concat"]], 1)
    t = torch.arange(B) * 100 + 1
    e1, taps1, n1, pl = forward(model, "1", x, t, inp["c_crossattn"], monkeypatch, want_taps=True)
    assert n1 == 16, "every SpatialTransformer of the tiny UNet should run on the engine, got %d" % n1
    assert xcd_status(pl) == 0
    sync = pl.xcd_sync.cpu()
    assert int(sync.abs().sum()) == 0, "the engine must leave its synchronisation words zeroed"
    e1b = model.model.diffusion_model(x.cuda(), t.cuda(), context=inp["c_crossattn"].cuda())
    assert torch.equal(e1, e1b), "rerun of the engine path is not bit-identical"
    e0, taps0, n0, _ = forward(model, "0", x, t, inp["c_crossattn"], monkeypatch, want_taps=True)
    assert n0 == 0
    for k in taps0:
        a, b = taps1[k], taps0[k]
        err = float((a - b).abs().max()) / max(1.0, float(b.abs().max()))
        assert err < 3e-2, "block %s: engine vs launch chain differ by %.3g of the block's range" % (k, err)
    ref = o_unet.unet_forward(sd, synth.TINY_UNET, x, t, inp["c_crossattn"])
    assert mse(e1, ref) < 1e-4 and mse(e0, ref) < 1e-4


@pytest.mark.parametrize("kind", ["tiny", "bbox", "upscale"])
def test_engine_forward_vs_reference_golden(kind, monkeypatch):
    """The golden eps of the REAL reference (tests/golden/<kind>.npz, B = 2, t = [981, 401], latent 32x24) with every
    transformer the engine takes running on it (XCDs 2-7 idle)."""
    model, sd = get_model(kind, monkeypatch)
    g = np.load(os.path.join(G, kind + ".npz"))
    k = KIND[kind]
    inp = synth.synth_inputs(2, (32, 24), k["C"], k["ntok"], 768, seed=0, concat_channels=k["cc"], steps=10)
    x = torch.cat([inp["x_T"], inp["c_concat"]], 1)
    eps, _, n_x, pl = forward(model, "1", x, torch.tensor([981, 401]), inp["c_crossattn"], monkeypatch)
    assert n_x >= 1, "no transformer of the %s UNet ran on the engine" % kind
    assert xcd_status(pl) == 0
    e = mse(eps, g["unet_eps"])
    assert e < 1e-4, "eps MSE vs reference golden %g with %d transformers on the engine" % (e, n_x)


def test_bench_shape_auto_mode_matches_the_chain(monkeypatch):
    """B = 8, latent 32x32 (BASELINE configs[1]): UPGPT_XCD=auto routes the 16x16 / 8x8 / 4x4 transformers to the engine;
    eps against the launch chain (the default) and the status word."""
    model, sd = get_model("bbox", monkeypatch)
    inp = synth.synth_inputs(8, (32, 32), 4, 87, 768, seed=0, text_only=True)
    x = torch.cat([inp["x_T"], inp["c_concat"]], 1)
    t = torch.full((8,), 981)
    ea, _, na, pl = forward(model, "auto", x, t, inp["c_crossattn"], monkeypatch)
    assert na >= 11 and xcd_status(pl) == 0
    e0, _, n0, _ = forward(model, "0", x, t, inp["c_crossattn"], monkeypatch)
    assert n0 == 0
    assert mse(ea, e0) < 2e-5, mse(ea, e0)
    assert float((ea.cpu() - e0.cpu()).abs().max()) < 3e-2 * max(1.0, float(e0.abs().max()))


def test_phase_check_refuses_bad_descriptors():
    from upgpt_amd import _lib as L
    ctx = L.get_context()
    q = L.XPhase()
    q.kind, q.n = L.XP_GEMM, 64
    assert ctx.lib.upk_xcd_phase_check(ctx.h, C.byref(q)) != 0  # null tensors
    buf = torch.zeros(1 << 16, dtype=torch.float16, device="cuda")
    q.a = q.w = q.y = buf.data_ptr()
    q.lda, q.k1, q.ntiles, q.n_out, q.ldy, q.pm, q.pn, q.mb, q.tn = 64, 48, 4, 64, 64, 1, 4, 64, 1
    assert ctx.lib.upk_xcd_phase_check(ctx.h, C.byref(q)) == L_ESHAPE()  # K not a multiple of 32
    q.k1 = 64
    assert ctx.lib.upk_xcd_phase_check(ctx.h, C.byref(q)) == 0
    q.mb = 80
    assert ctx.lib.upk_xcd_phase_check(ctx.h, C.byref(q)) == L_ESHAPE()
    q.mb, q.pm, q.pn = 64, 8, 8
    assert ctx.lib.upk_xcd_phase_check(ctx.h, C.byref(q)) == L_ESHAPE()  # 64 CUs on an XCD of 32
    q.pm, q.pn, q.kind = 1, 4, 9
    assert ctx.lib.upk_xcd_phase_check(ctx.h, C.byref(q)) != 0


def L_ESHAPE():
    return -2

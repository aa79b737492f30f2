"""CLIP image tower of the style-conditioning stage on the HIP kernels (second half).

The reference's `FrozenClipImageEmbedder2` (ldm/modules/encoders/modules.py:234-256) holds OpenAI's `clip` package model
`ViT-L/14` (a third-party dependency, git main, weights downloaded at run time) and calls `model.encode_image` on
[b, n, 3, 224, 224] pre-processed crops -> [b, n, 768].  `clip.model.VisionTransformer`: Conv2d(3, 1024, 14, stride 14,
bias=False) patch embedding -> [class token ; 256 patches] + positional embedding -> ln_pre -> 24 pre-LN residual blocks
(16 heads of 64, fused in_proj q|k|v, MLP 1024 -> 4096 -> 1024 with QuickGELU) -> ln_post on the class token -> @ proj
[1024, 768].

* `CLIPVisual` holds the weights under the package's own names (`conv1.weight`, `class_embedding`,
  `positional_embedding`, `transformer.resblocks.N.attn.in_proj_weight`, `...mlp.c_fc.weight`, `ln_post.*`, `proj`), so a
  checkpoint's `extra_cond_models.*.model.visual.*` entries load with `load_state_dict` (the text half of the CLIP model
  is never used by this stage and is not held).
* Compute: `upk_patchify_nchw_f32_f16` + GEMM, `upk_vit_assemble_f16`, `upk_layernorm_f16` (ln_pre / ln_post), per block
  LayerNorm folded into the q|k|v projection and into c_fc (`upk_conv_desc.ln_colsum`), `upk_attention_f16`,
  `UPK_F_QUICKGELU`, residual epilogues; the final projection is a GEMM over the class-token rows.
"""
import os

import torch
from torch import nn

from . import _lib as L
from .engine import Act, Emitter, PW, Packer, Program, _rup
from .packing import SharedPacks
from .params import ParamTree, weights_fingerprint

CLIP_L14_VISION = dict(width=1024, layers=24, heads=16, patch_size=14, image_size=224, output_dim=768, eps=1e-5)


def visual_param_shapes(cfg):
    d, p = cfg["width"], cfg["patch_size"]
    n = (cfg["image_size"] // p) ** 2 + 1
    s = {"class_embedding": (d,), "positional_embedding": (n, d), "proj": (d, cfg["output_dim"]),
         "conv1.weight": (d, 3, p, p), "ln_pre.weight": (d,), "ln_pre.bias": (d,), "ln_post.weight": (d,),
         "ln_post.bias": (d,)}
    for i in range(cfg["layers"]):
        b = "transformer.resblocks.%d." % i
        s[b + "attn.in_proj_weight"], s[b + "attn.in_proj_bias"] = (3 * d, d), (3 * d,)
        s[b + "attn.out_proj.weight"], s[b + "attn.out_proj.bias"] = (d, d), (d,)
        s[b + "ln_1.weight"], s[b + "ln_1.bias"] = (d,), (d,)
        s[b + "mlp.c_fc.weight"], s[b + "mlp.c_fc.bias"] = (4 * d, d), (4 * d,)
        s[b + "mlp.c_proj.weight"], s[b + "mlp.c_proj.bias"] = (d, 4 * d), (d,)
        s[b + "ln_2.weight"], s[b + "ln_2.bias"] = (d,), (d,)
    return s


class _VisualPlan(Emitter):
    """Launch program of the image tower for N images."""

    def __init__(self, ctx, cfg, get, N, shared=None):
        super().__init__(ctx)
        self.cfg, self.N = cfg, N
        d, heads, p, img = cfg["width"], cfg["heads"], cfg["patch_size"], cfg["image_size"]
        dh = d // heads
        if dh not in (32, 64, 128) or d % 32:
            raise NotImplementedError("CLIP image tower with head dim %d / width %d" % (dh, d))
        eps = float(cfg["eps"])
        g = img // p
        npatch, S = g * g, g * g + 1
        kp = _rup(3 * p * p, 32)
        chk, h = self._chk, self.hctx
        P = self.prog = Program(ctx)

        # names in the packer's "<name>.weight / .bias" convention on top of the OpenAI keys
        def lookup(n):
            if n.endswith("attn.in_proj.weight"):
                return get(n.replace("in_proj.weight", "in_proj_weight"))
            if n.endswith("attn.in_proj.bias"):
                return get(n.replace("in_proj.bias", "in_proj_bias"))
            if n == "conv1_flat.weight":
                return get("conv1.weight").reshape(d, 3 * p * p)
            if n == "proj_t.weight":
                return get("proj").t().contiguous()
            return get(n)

        pk = Packer(ctx, lookup)
        if shared is not None:  # (the packed tower is shared by every lane's plan: 0.6 GB once, not once per lane)
            pk = shared.wrap(pk)
        self.x = self.alloc(N, 3, img, img, dtype=torch.float32)
        patches = Act(self.alloc(N * npatch, kp), N, npatch, 1, kp)
        fn_p = self.lib.upk_patchify_nchw_f32_f16
        ap = (self.x.data_ptr(), N, 3, img, img, p, patches.t.data_ptr(), kp)
        P.add(lambda s: chk(fn_p(h, *ap, s)), self.x, patches, cls="other")
        pe = self.conv(P, patches, pk.pack("conv1_flat", cin_packed=kp, bias=False))
        cls_e, pos_e = pk.vec("class_embedding"), pk.vec("positional_embedding")
        tok = Act(self.alloc(N * S, d), N, S, 1, d)
        fn_v = self.lib.upk_vit_assemble_f16
        av = (pe.t.data_ptr(), pe.ld, cls_e.data_ptr(), pos_e.data_ptr(), N, npatch, d, tok.t.data_ptr(), tok.ld)
        P.add(lambda s: chk(fn_v(h, *av, s)), pe, cls_e, pos_e, tok, cls="other")
        x = Act(self.alloc(N * S, d), N, S, 1, d)
        self._ln(P, tok, pk.vec("ln_pre.weight"), pk.vec("ln_pre.bias"), eps, x, N * S)
        vt_ld = _rup(S, 32)
        fn_a = self.lib.upk_attention_f16
        M = N * S
        for i in range(cfg["layers"]):
            b = "transformer.resblocks.%d." % i
            wqkv = pk.pack(b + "attn.in_proj", n_out=2 * d, ln=b + "ln_1")  # rows already q | k | v
            qk = Act(self.alloc(M, 2 * d), N, S, 1, 2 * d)
            vt = self.alloc(N, heads, dh, vt_ld, zero=True)
            self.conv(P, x, wqkv, out=qk, ln_eps=eps,
                      vt=dict(t=vt, heads=heads, dhead=dh, ld=vt_ld, tokens=S, **{"from": 2 * d}))
            att = Act(self.alloc(M, d), N, S, 1, d)
            aa = (qk.t.data_ptr(), 2 * d, S * 2 * d, qk.t[:, d:].data_ptr(), 2 * d, S * 2 * d, vt.data_ptr(), vt_ld,
                  att.t.data_ptr(), d, S * d, N, heads, S, S, dh, float(dh ** -0.5))
            P.add(lambda s, aa=aa: chk(fn_a(h, *aa, s)), qk, vt, att, cls="attention")
            x = self.conv(P, att, pk.pack(b + "attn.out_proj"), residual=x)
            hmid = self.conv(P, x, pk.pack(b + "mlp.c_fc", ln=b + "ln_2"), ln_eps=eps, flags=L.F_QUICKGELU)
            x = self.conv(P, hmid, pk.pack(b + "mlp.c_proj"), residual=x)
        # ln_post on the class tokens (row n*S of x: a [N, d] view with leading dimension S*d), then @ proj
        cls_rows = Act(self.alloc(N, d), N, 1, 1, d)
        self._ln(P, x, pk.vec("ln_post.weight"), pk.vec("ln_post.bias"), eps, cls_rows, N, ldx=S * x.ld)
        self.out = self.alloc(N, cfg["output_dim"], dtype=torch.float32)
        self.conv(P, cls_rows, pk.pack("proj_t", bias=False), out_f32=self.out)
        self.apply_tuning(tune_missing=os.environ.get("UPGPT_AUTOTUNE", "0") == "1")

    def _ln(self, P, x, gamma, beta, eps, y, rows, ldx=None):
        fn, h, chk = self.lib.upk_layernorm_f16, self.hctx, self._chk
        a = (x.t.data_ptr(), x.ld if ldx is None else ldx, rows, x.C, gamma.data_ptr(), beta.data_ptr(), eps,
             y.t.data_ptr(), y.ld)
        P.add(lambda s: chk(fn(h, *a, s)), x, gamma, beta, y, cls="layernorm")

    def upload(self, images):
        """The only host -> device step of the tower (bracketed by _lib.host_io when the crops come from the host)."""
        img = self.cfg["image_size"]
        if tuple(images.shape) != (self.N, 3, img, img):
            raise ValueError("images must be [%d, 3, %d, %d], got %s" % (self.N, img, img, tuple(images.shape)))
        self.x.copy_(images.to(self.dev, torch.float32))

    def execute(self):
        self.prog.run()
        return self.out.clone()

    def run(self, images):
        self.upload(images)
        return self.execute()


class CLIPVisual(ParamTree):
    """`clip.model.VisionTransformer` weights + forward ([N, 3, 224, 224] -> [N, 768])."""

    def __init__(self, **config):
        super().__init__()
        self.config = dict(CLIP_L14_VISION, **config)
        self.add_params(visual_param_shapes(self.config))
        self._plans = {}
        self._fp = None

    def forward(self, images):
        p = next(self.parameters())
        if p.device.type != "cuda":
            raise RuntimeError("upgpt_amd.CLIPVisual computes only through the HIP kernels on an MI355X: move it to 'cuda' "
                               "first. There is no CPU fallback.")
        from ._lib import PLAN_LOCK, concurrency, current_lane, get_context, host_io
        with PLAN_LOCK:  # (execution lanes: a plan — buffers and packed weights — per (crop count, tuning table, lane))
            fp = weights_fingerprint(self)
            if fp != self._fp:
                self._plans, self._fp, self._packs = {}, fp, SharedPacks()
            N = int(images.shape[0])
            key = (N, concurrency() > 1, current_lane())
            plan = self._plans.get(key)
            if plan is None:
                mine = [k for k in self._plans if k[-1] == key[-1]]
                if len(mine) >= 2:
                    self._plans.pop(mine[0])
                params = dict(self.named_parameters())
                with torch.cuda.device(p.device), host_io():
                    plan = self._plans[key] = _VisualPlan(get_context(p.device), self.config, lambda n: params[n].data, N,
                                                          shared=self._packs)
                    if self._packs.take_fresh():
                        torch.cuda.current_stream(p.device).synchronize()  # (packed on this lane's stream, read from every lane's)
        import contextlib
        from ._lib import host_io
        with torch.cuda.device(p.device), host_io():
            # the whole tower stays serialised against the other lanes' graph captures: with only the upload under the lock
            #  the secondary bench hit "capturing stream has unjoined work" once in five sessions — a tower's first
            # eager launches set function attributes and allocate while another lane captures.  Packed weights are shared.
            plan.upload(images)
            return plan.execute()


class _CLIPModelShell(nn.Module):
    """The part of `clip.model.CLIP` this stage touches: `.visual` and `.encode_image`."""

    def __init__(self, **config):
        super().__init__()
        self.visual = CLIPVisual(**config)

    def encode_image(self, image):
        return self.visual(image)


class FrozenClipImageEmbedder2(nn.Module):
    """Drop-in for ldm.modules.encoders.modules.FrozenClipImageEmbedder2 (modules.py:234-256): [b, n, 3, 224, 224]
    pre-processed crops -> [b, n, 768]; state-dict keys `model.visual.*` as in the reference's checkpoints."""

    def __init__(self, model="ViT-L/14", jit=False, device="cuda", **config):
        super().__init__()
        if model != "ViT-L/14" and not config:
            raise NotImplementedError("only the ViT-L/14 image tower of the reference configs is described here; pass the "
                                      "VisionTransformer dimensions as keyword arguments for another one")
        self.model = _CLIPModelShell(**config).eval()
        for p in self.parameters():
            p.requires_grad = False

    @torch.no_grad()
    def forward(self, x):
        b, n, c, h, w = x.shape
        ret = self.model.encode_image(x.reshape(b * n, c, h, w))
        return ret.reshape(b, n, -1)

    def encode(self, x):
        return self(x)

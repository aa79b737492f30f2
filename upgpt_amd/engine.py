"""Lowers arch.* records + fp32 master weights to flat programs of libupk.so launches.

Design (MI355X-first, not a translation of the reference's nn.Module.forward chain):
  * activations live as token-major NHWC fp16 [B*H*W, C] for the whole network, so
    conv1x1 == Linear == GEMM and the reference's `b c h w <-> b (h w) c` rearranges
    (attention.py:256,259) and head split/merge copies (:178,192) do not exist;
  * channel concats (openaimodel.py:736) are never materialised — GroupNorm and the
    implicit-GEMM kernel read two sources; nearest-2x upsampling is folded into the
    following conv's addressing; bias / timestep-embedding add / residual / GEGLU / SiLU
    are GEMM epilogues;
  * shapes are static per (B, H, W, n_ctx): every buffer is allocated once, every launch
    descriptor is built once, and the per-step program is captured into ONE HIP graph that
    the sampler replays S times; the step index lives on the device (no per-step host
    work, no torch.full allocations as in ddim.py:189-192);
  * step-invariant work is hoisted out of the loop: the timestep-embedding MLP and all 22
    emb_layers projections for ALL S steps (one batched GEMM each) and the cross-attention
    K / V projections of the context.

PyTorch here = device allocator + stream handle only.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from . import knobs as K
from ._check import require
from .arch import UNetArch, VAEArch
from .emitter import Act, Emitter, Program  # noqa: F401  (re-exported: scripts and tests import them from here)
from .packing import (PW, PackedUNet, PackedVAEDecoder, PackedVAEEncoder, Packer, _rup, geglu_rows_map,  # noqa: F401
                      head_pad, pad_rows_map, qproj_pack)
from .tuning import TUNE_CACHE, TuneCache  # noqa: F401


class UNetPlan(Emitter):
    """Buffers + programs of one UNet for fixed (B, H, W, n_ctx, rows).

    rows = number of timestep-embedding rows: B in 'forward' mode (per-sample t, as
    UNetModel.forward takes them), S in 'sampler' mode (one row per DDIM step, shared by
    the batch: ddim.py:142 uses the same t for every sample)."""

    def __init__(self, ctx, packed: PackedUNet, B, H, W, n_ctx, rows, mode):
        super().__init__(ctx)
        require(mode in ("forward", "sampler"), "UNetPlan mode must be 'forward' or 'sampler'", ValueError)
        self.pk, self.arch = packed, packed.arch
        self.B, self.H, self.W, self.n_ctx, self.rows, self.mode = B, H, W, n_ctx, rows, mode
        a = self.arch
        mc, te = a.model_channels, a.time_embed_dim
        self.cin_pad = _rup(a.in_channels, 32)
        # inputs / outputs
        self.xin = Act(self.alloc(B * H * W, self.cin_pad, zero=True), B, H, W, a.in_channels)
        self.ctx32 = self.alloc(B * n_ctx, a.context_dim, dtype=torch.float32)
        self.ctx16 = Act(self.alloc(B * n_ctx, a.context_dim), 1, B * n_ctx, 1, a.context_dim)
        self.t_rows = self.alloc(rows, dtype=torch.float32)
        self.eps = self.alloc(B, a.out_channels, H, W, dtype=torch.float32)
        self.step = self.alloc(1, dtype=torch.int32, zero=True)
        self.gn_ws = self.alloc(max(64, ctx.groupnorm_ws_bytes(B, H * W) // 4), dtype=torch.float32)
        self.rowvecs = {}
        self.kv = {}
        self.prep = Program(ctx)
        self.body = Program(ctx)
        self._emit_prep()
        self._emit_body()
        self.n_prefetch_links = self.link_weight_prefetch(self.body)

    # ---- step-invariant part
    def _emit_prep(self):
        P, a, w = self.prep, self.arch, self.pk.w
        mc, te, R = a.model_channels, a.time_embed_dim, self.rows
        lib, h, chk = self.lib, self.hctx, self._chk
        c32, c16 = self.ctx32, self.ctx16
        P.add(lambda s: chk(lib.upk_f32_to_f16(h, c32.data_ptr(), c32.shape[0], c32.shape[1], c16.t.data_ptr(),
                                               c16.ld, s)))
        temb = Act(self.alloc(R, mc), 1, R, 1, mc)
        tr = self.t_rows
        P.add(lambda s: chk(lib.upk_timestep_embed_f16(h, tr.data_ptr(), R, mc, 10000.0, temb.t.data_ptr(), mc, s)))
        # emb = Linear(SiLU(Linear(temb))) (openaimodel.py:506-511); every consumer applies SiLU
        # first (openaimodel.py:219), so SiLU(emb) is what is kept.
        h1 = self.conv(P, temb, w["time_embed.0"], flags=L.F_SILU)
        semb = self.conv(P, h1, w["time_embed.2"], flags=L.F_SILU)
        for Lr in a.all_layers():
            if Lr.kind == "res":
                rv = self.alloc(R, Lr.cout, dtype=torch.float32)
                self.conv(P, semb, w[Lr.name + ".emb_layers.1"], out_f32=rv)
                self.rowvecs[Lr.name] = rv
            elif Lr.kind == "st":
                heads, dp = Lr.heads, head_pad(Lr.dhead)
                hd = heads * dp
                vt_ld = _rup(self.n_ctx, 32)
                kc = Act(self.alloc(self.B * self.n_ctx, hd), 1, self.B * self.n_ctx, 1, hd)
                vtc = self.alloc(self.B, heads, dp, vt_ld, zero=True)
                self.conv(P, self.ctx16, w[Lr.name + ".transformer_blocks.0.attn2.kv"], out=kc,
                          vt=dict(t=vtc, heads=heads, dhead=dp, ld=vt_ld, tokens=self.n_ctx, **{"from": hd}))
                self.kv[Lr.name] = (kc, vtc, vt_ld)

    # ---- per-step part
    def _rv(self, name):
        rv = self.rowvecs[name]
        if self.mode == "sampler":
            return dict(rowvec=rv, rv_bs=0, rv_ss=rv.shape[1], step=self.step)
        return dict(rowvec=rv, rv_bs=rv.shape[1], rv_ss=0, step=None)

    def _res(self, P, Lr, x, skip):
        w, v = self.pk.w, self.pk.v
        n = Lr.name
        if skip is not None and (skip.H, skip.W, skip.B) != (x.H, x.W, x.B):
            # the reference fails in torch.cat here (openaimodel.py:736) when H or W is not a
            # multiple of 2^(levels-1); fail just as loudly instead of reading mismatched rows
            raise ValueError("UNet skip connection %dx%d does not match the decoder feature map %dx%d at %s: "
                             "latent height and width must be multiples of %d" % (
                                 skip.H, skip.W, x.H, x.W, n, 2 ** (len(self.arch.channel_mult) - 1)))
        g1, b1 = v[n + ".in_layers.0"]
        hh = self.conv(P, x, w[n + ".in_layers.2"], x2=skip, gn=(g1, b1, 1e-5, True, self.gn_ws), gn_stats=True,
                       **self._rv(n))
        gn2 = (*v[n + ".out_layers.0"], 1e-5, True, self.gn_ws, True)  # (hh has no reader but this GroupNorm)
        if Lr.cin != Lr.cout:
            if self.fold_skip(hh, w[n + ".out_layers.3"], w[n + ".skip_connection"], x, skip):
                # skip projection as an appended K segment of the second conv: one launch, no residual round trip
                return self.conv(P, hh, w[n + ".out_layers.3+skip"], append=(x, skip), gn=gn2, gn_stats=True)
            sk = self.conv(P, x, w[n + ".skip_connection"], x2=skip)
        else:
            require(skip is None, "decoder ResBlock without channel change got a skip tensor", RuntimeError)
            sk = x
        return self.conv(P, hh, w[n + ".out_layers.3"], residual=sk, gn=gn2, gn_stats=True)

    def _st(self, P, Lr, x):
        w, v = self.pk.w, self.pk.v
        n = Lr.name
        t = n + ".transformer_blocks.0"
        B, HW, M = x.B, x.H * x.W, x.M
        heads, dh = Lr.heads, Lr.dhead
        dp = head_pad(dh)
        hd = heads * dp
        scale = dh ** -0.5
        # self-attention
        qk = Act(self.alloc(M, 2 * hd), B, x.H, x.W, 2 * hd)
        vt_ld = _rup(HW, 32)
        vt = self.alloc(B, heads, dp, vt_ld, zero=True)
        t0 = None
        if self.head_block_ok(x, t, heads, dp, qk, vt_ld):
            t0 = self.head_block(P, x, n, t, heads, dp, qk, vt, vt_ld, (*v[n + ".norm"], 1e-6, self.gn_ws))
        if t0 is None:
            t0 = self.conv(P, x, w[n + ".proj_in"], gn=(*v[n + ".norm"], 1e-6, False, self.gn_ws))
            self.ln_linear(P, t0, t + ".attn1.qkv", t + ".norm1", out=qk,  # norm1 -> q|k|v (attention.py:203,212)
                           vt=dict(t=vt, heads=heads, dhead=dp, ld=vt_ld, tokens=HW, **{"from": 2 * hd}))
        a1 = Act(self.alloc(M, hd), B, x.H, x.W, hd)
        self.attention(P, qk.t, 2 * hd, HW * 2 * hd, qk.t[:, hd:], 2 * hd, HW * 2 * hd, vt, vt_ld, a1.t, hd, HW * hd,
                       B, heads, HW, HW, dp, scale)
        P.attn_flops += 4 * B * heads * HW * HW * dh
        # cross-attention over the (precomputed) context K / V
        kc, vtc, cld = self.kv[n]
        t2 = self.cross_block(P, a1, t0, t, kc, vtc, cld, heads, dp, scale)
        if t2 is not None:
            return self._st_ff(P, Lr, x, t2)
        t1 = self.conv(P, a1, w[t + ".attn1.to_out"], residual=t0)
        a2 = Act(self.alloc(M, hd), B, x.H, x.W, hd)
        # (pays while there are >= 2 waves per SIMD to hide a wave's serial projection -> scores chain: 32x32 level
        # 16 -> 12 us per block; at 16x16 (1 wave per SIMD, 4x the weight slice per wave) 16 -> 18 us)
        if (K.QPROJ_FUSE and (t + ".attn2.qproj") in w and t1.C == t1.ld and dp == 32
                and B * heads * ((HW + 31) // 32) >= 2048):
            # norm2 -> to_q inside the attention kernel (include/upk.h upk_attention_qproj_f16): no q GEMM, no q tensor
            wq, wu, wb = w[t + ".attn2.qproj"]
            fn, hh, chk = self.lib.upk_attention_qproj_f16, self.hctx, self._chk
            qa = (t1.t.data_ptr(), t1.ld, HW * t1.ld, t1.C, t1.C, 1e-5, wq.data_ptr(), wu.data_ptr(), wb.data_ptr(),
                  kc.t.data_ptr(), hd, self.n_ctx * hd, vtc.data_ptr(), cld, a2.t.data_ptr(), hd, HW * hd, B, heads, HW,
                  self.n_ctx, dp, float(scale))
            P.add(lambda s: chk(fn(hh, *qa, s)), t1, wq, wu, wb, kc, vtc, a2, cls="attention",
                  label="attn+q B%d h%d nq%d nkv%d d%d C%d" % (B, heads, HW, self.n_ctx, dp, t1.C))
            P.igemm_flops += 2 * M * heads * dh * t1.C
        else:
            q2 = self.ln_linear(P, t1, t + ".attn2.q", t + ".norm2")
            self.attention(P, q2.t, hd, HW * hd, kc.t, hd, self.n_ctx * hd, vtc, cld, a2.t, hd, HW * hd, B, heads, HW,
                           self.n_ctx, dp, scale)
        P.attn_flops += 4 * B * heads * HW * self.n_ctx * dh
        t2 = self.conv(P, a2, w[t + ".attn2.to_out"], residual=t1)
        return self._st_ff(P, Lr, x, t2)

    def _st_ff(self, P, Lr, x, t2):
        """GEGLU feed-forward of the block + the transformer's proj_out (attention.py:261, 330-336)."""
        w = self.pk.w
        n = Lr.name
        t = n + ".transformer_blocks.0"
        fused = self.geglu_mlp(P, t2, x, w[t + ".ff.geglu_ln"], w[n + ".ff.out+proj_out"])
        if fused is not None:
            return fused
        ff = self.ln_linear(P, t2, t + ".ff.geglu", t + ".norm3", flags=L.F_GEGLU)
        if self.fold_ff_out(ff, t2, w[t + ".ff.out"]):
            return self.conv(P, ff, w[n + ".ff.out+proj_out"], residual=x, append=(t2, None), gn_stats=True)
        t3 = self.conv(P, ff, w[t + ".ff.out"], residual=t2)
        return self.conv(P, t3, w[n + ".proj_out"], residual=x, gn_stats=True)

    def _layers(self, P, layers, x, skip=None):
        w = self.pk.w
        for Lr in layers:
            if Lr.kind == "conv":
                x = self.conv(P, x, w[Lr.name], gn_stats=True)
            elif Lr.kind == "res":
                x = self._res(P, Lr, x, skip)
                skip = None
            elif Lr.kind == "st":
                x = self._st(P, Lr, x)
            elif Lr.kind == "down":
                x = self.conv(P, x, w[Lr.name + ".op"], stride=2, gn_stats=True)
            elif Lr.kind == "up":
                x = self.conv(P, x, w[Lr.name + ".conv"], flags=L.F_UPSAMPLE2X, gn_stats=True)
        return x

    def _emit_body(self):
        P, a = self.body, self.arch
        hs = []
        x = self.xin
        self.taps = {}  # block name -> Act (debug / parity tests)
        for i, blk in enumerate(a.input_blocks):
            x = self._layers(P, blk, x)
            hs.append(x)
            self.taps["input_blocks.%d" % i] = x
        x = self._layers(P, a.middle_block, x)
        self.taps["middle_block"] = x
        for i, blk in enumerate(a.output_blocks):
            x = self._layers(P, blk, x, skip=hs.pop())
            self.taps["output_blocks.%d" % i] = x
        self.conv(P, x, self.pk.w["out.2"], nchw_out=self.eps, gn=(*self.pk.v["out.0"], 1e-5, True, self.gn_ws))

    # ---- host-facing helpers
    def close(self):
        """Releases the graphs of the sampler states attached to this plan (ddim.py / plms.py fast paths)."""
        for attr in ("_sampler_state", "_sampler_state_cfg", "_plms_state", "_plms_state_cfg"):
            st = self.__dict__.pop(attr, None)
            if st is not None:
                st.close()

    def load_context(self, context):
        """context: [B, n_ctx, context_dim] tensor (any float dtype / device)."""
        require(tuple(context.shape) == (self.B, self.n_ctx, self.arch.context_dim), lambda: "context shape %s != plan %s" % (tuple(context.shape), (self.B, self.n_ctx, self.arch.context_dim)), ValueError)
        self.ctx32.copy_(context.reshape(self.B * self.n_ctx, -1).to(self.dev, torch.float32), non_blocking=True)

    def load_x_nchw(self, x, c_off=0, zero_pad_to=0):
        """fp32 NCHW [B, c, H, W] -> channels [c_off, c_off + c) of the stem input."""
        x = x.to(self.dev, torch.float32).contiguous()
        require(x.shape[0] == self.B and tuple(x.shape[2:]) == (self.H, self.W), lambda: "stem input %s does not match the plan (B=%d, %dx%d)" % (tuple(x.shape), self.B, self.H, self.W), ValueError)
        self.ctx.nchw_to_nhwc(x, self.B, x.shape[1], self.H * self.W, self.xin.t, self.xin.ld, c_off, zero_pad_to, 1.0)


class VAEDecodePlan(Emitter):
    """decode_first_stage (ddpm.py:771-829 plain branch): z / scale_factor ->
    post_quant_conv -> Decoder (model.py:535-568) -> fp32 NCHW image."""

    def __init__(self, ctx, packed: PackedVAEDecoder, B, h, w, scale_factor):
        super().__init__(ctx)
        self.pk, self.arch = packed, packed.arch
        a = self.arch
        self.B, self.h, self.w = B, h, w
        f = a.factor
        self.z = self.alloc(B, a.embed_dim, h, w, dtype=torch.float32)
        self.img = self.alloc(B, a.out_ch, h * f, w * f, dtype=torch.float32)
        self.gn_ws = self.alloc(max(64, ctx.groupnorm_ws_bytes(B, h * f * w * f) // 4), dtype=torch.float32)
        self.prog = Program(ctx)
        P, W_, V_ = self.prog, packed.w, packed.v
        zin = Act(self.alloc(B * h * w, _rup(a.embed_dim, 32), zero=True), B, h, w, a.embed_dim)
        lib, hh, chk, z = self.lib, self.hctx, self._chk, self.z
        inv = 1.0 / float(scale_factor)
        P.add(lambda s: chk(lib.upk_nchw_f32_to_nhwc_f16(hh, z.data_ptr(), B, a.embed_dim, h * w, zin.t.data_ptr(),
                                                         zin.ld, 0, 0, inv, s)))
        x = self.conv(P, zin, W_["post_quant_conv"],
                      out=Act(self.alloc(B * h * w, _rup(a.z_channels, 32), zero=True), B, h, w, a.z_channels))
        for Lr in a.decoder:
            n = Lr.name
            if Lr.kind == "conv":
                x = self.conv(P, x, W_[n], gn_stats=K.vae_gn_byproduct(x.M))
            elif Lr.kind == "resnet":
                h1 = self.conv(P, x, W_[n + ".conv1"], gn=(*V_[n + ".norm1"], 1e-6, True, self.gn_ws), gn_stats=K.vae_gn_byproduct(x.M))
                sk = self.conv(P, x, W_[n + ".nin_shortcut"]) if Lr.cin != Lr.cout else x
                x = self.conv(P, h1, W_[n + ".conv2"], residual=sk, gn=(*V_[n + ".norm2"], 1e-6, True, self.gn_ws),
                              gn_stats=K.vae_gn_byproduct(x.M))
            elif Lr.kind == "attn":
                c, HW = Lr.ch, x.H * x.W
                xn = self.groupnorm(P, x, *V_[n + ".norm"], 1e-6, False, self.gn_ws)
                qk = Act(self.alloc(x.M, 2 * c), x.B, x.H, x.W, 2 * c)
                vt_ld = _rup(HW, 32)
                vt = self.alloc(x.B, 1, c, vt_ld, zero=True)
                self.conv(P, xn, W_[n + ".qkv"], out=qk,
                          vt=dict(t=vt, heads=1, dhead=c, ld=vt_ld, tokens=HW, **{"from": 2 * c}))
                ao = Act(self.alloc(x.M, c), x.B, x.H, x.W, c)
                self.attention(P, qk.t, 2 * c, HW * 2 * c, qk.t[:, c:], 2 * c, HW * 2 * c, vt, vt_ld, ao.t, c, HW * c,
                               x.B, 1, HW, HW, c, int(c) ** -0.5)
                x = self.conv(P, ao, W_[n + ".proj_out"], residual=x, gn_stats=K.vae_gn_byproduct(x.M))
            elif Lr.kind == "upconv":
                x = self.conv(P, x, W_[n], flags=L.F_UPSAMPLE2X)
            elif Lr.kind == "norm_out":
                x = self.groupnorm(P, x, *V_[n], 1e-6, True, self.gn_ws)
            elif Lr.kind == "conv_out":
                self.conv(P, x, W_[n], nchw_out=self.img)
        self.graph = None

    def run(self, z):
        z = z.to(self.dev, torch.float32).contiguous()
        require(tuple(z.shape) == tuple(self.z.shape), lambda: repr((z.shape, self.z.shape)), ValueError)
        self.z.copy_(z)
        self.prog.run()
        return self.img


class VAEEncodePlan(Emitter):
    """AutoencoderKL.encode (autoencoder.py:324-328): Encoder (model.py:434-459) -> quant_conv ->
    posterior moments [B, 2*embed_dim, h, w] fp32 NCHW.  Same kernels as the decoder; the
    stride-2 Downsample uses the asymmetric (0,1,0,1) padding mode of the implicit GEMM."""

    def __init__(self, ctx, packed: PackedVAEEncoder, B, H, W):
        super().__init__(ctx)
        self.pk, self.arch = packed, packed.arch
        a = self.arch
        f = a.factor
        require(H % f == 0 and W % f == 0, lambda: "image size must be a multiple of %d" % f, ValueError)
        self.x = self.alloc(B, a.in_channels, H, W, dtype=torch.float32)
        self.moments = self.alloc(B, 2 * a.embed_dim, H // f, W // f, dtype=torch.float32)
        self.gn_ws = self.alloc(max(64, ctx.groupnorm_ws_bytes(B, H * W) // 4), dtype=torch.float32)
        self.prog = Program(ctx)
        P, W_, V_ = self.prog, packed.w, packed.v
        xin = Act(self.alloc(B * H * W, _rup(a.in_channels, 32), zero=True), B, H, W, a.in_channels)
        lib, hh, chk, xs = self.lib, self.hctx, self._chk, self.x
        P.add(lambda s: chk(lib.upk_nchw_f32_to_nhwc_f16(hh, xs.data_ptr(), B, a.in_channels, H * W,
                                                         xin.t.data_ptr(), xin.ld, 0, 0, 1.0, s)))
        x = xin
        for Lr in a.encoder:
            n = Lr.name
            if Lr.kind == "conv":
                x = self.conv(P, x, W_[n], gn_stats=K.vae_gn_byproduct(x.M))
            elif Lr.kind == "resnet":
                h1 = self.conv(P, x, W_[n + ".conv1"], gn=(*V_[n + ".norm1"], 1e-6, True, self.gn_ws), gn_stats=K.vae_gn_byproduct(x.M))
                sk = self.conv(P, x, W_[n + ".nin_shortcut"]) if Lr.cin != Lr.cout else x
                x = self.conv(P, h1, W_[n + ".conv2"], residual=sk, gn=(*V_[n + ".norm2"], 1e-6, True, self.gn_ws),
                              gn_stats=K.vae_gn_byproduct(x.M))
            elif Lr.kind == "attn":
                c, HW = Lr.ch, x.H * x.W
                xn = self.groupnorm(P, x, *V_[n + ".norm"], 1e-6, False, self.gn_ws)
                qk = Act(self.alloc(x.M, 2 * c), x.B, x.H, x.W, 2 * c)
                vt_ld = _rup(HW, 32)
                vt = self.alloc(x.B, 1, c, vt_ld, zero=True)
                self.conv(P, xn, W_[n + ".qkv"], out=qk,
                          vt=dict(t=vt, heads=1, dhead=c, ld=vt_ld, tokens=HW, **{"from": 2 * c}))
                ao = Act(self.alloc(x.M, c), x.B, x.H, x.W, c)
                self.attention(P, qk.t, 2 * c, HW * 2 * c, qk.t[:, c:], 2 * c, HW * 2 * c, vt, vt_ld, ao.t, c, HW * c,
                               x.B, 1, HW, HW, c, int(c) ** -0.5)
                x = self.conv(P, ao, W_[n + ".proj_out"], residual=x)
            elif Lr.kind == "downconv":
                x = self.conv(P, x, W_[n], stride=2, flags=L.F_PAD_ASYM)
            elif Lr.kind == "norm_out":
                x = self.groupnorm(P, x, *V_[n], 1e-6, True, self.gn_ws)
            elif Lr.kind == "conv_out":
                x = self.conv(P, x, W_[n], out=Act(self.alloc(x.M, _rup(Lr.cout, 32), zero=True), x.B, x.H, x.W,
                                                   Lr.cout))
        self.conv(P, x, W_["quant_conv"], nchw_out=self.moments)

    def run(self, img):
        img = img.to(self.dev, torch.float32).contiguous()
        require(tuple(img.shape) == tuple(self.x.shape), lambda: repr((img.shape, self.x.shape)), ValueError)
        self.x.copy_(img)
        self.prog.run()
        return self.moments


# ====================================================================== sampler step graph
class SamplerState:
    """Device state of one DDIM run on a sampler-mode UNetPlan: latent x (fp32 NCHW),
    pred_x0, the per-step coefficient / noise tables and the captured step graph
    (UNet body -> upk_ddim_step_f32 -> upk_advance_step)."""

    def __init__(self, plan: UNetPlan, channels, cfg=False, plms=False):
        """cfg: classifier-free guidance — the plan runs 2*B rows ([unconditional ; conditional], ddim.py:173-178),
        the latent state has B = plan.B // 2 samples and the update combines the two halves of eps.
        plms: the graph is one PLMS model evaluation (upk_plms_step_f32; plan rows = evaluations = steps + 1)."""
        require(plan.mode == "sampler", "SamplerState needs a sampler-mode plan", ValueError)
        self.plan = plan
        self.cfg = bool(cfg)
        self.plms = bool(plms)
        require(not cfg or plan.B % 2 == 0, "guidance runs [uncond ; cond]: the plan's batch must be even", ValueError)
        B, H, W, R = (plan.B // 2 if cfg else plan.B), plan.H, plan.W, plan.rows
        self.B = B
        self.C = channels
        self.x = plan.alloc(B, channels, H, W, dtype=torch.float32)
        self.pred_x0 = plan.alloc(B, channels, H, W, dtype=torch.float32)
        self.coefs = plan.alloc(R, 4, dtype=torch.float32)
        self.noise = None
        self.graphs = {}
        self.step_done = plan.alloc(1, dtype=torch.int32, zero=True)  # arrival counter of the step kernels
        self.hist = plan.alloc(3, B * channels * H * W, dtype=torch.float32) if plms else None

    def close(self):
        """Destroys the instantiated HIP graphs (they hold device memory; called when the owning plan is dropped).
        sample() hands back a clone enqueued behind the last replay, so a replay may still be in flight when a new shape
        evicts this plan: the device is drained first (HIP does not promise deferred destruction of an executing graph)."""
        if self.graphs:
            torch.cuda.synchronize(self.plan.dev)
        for g in self.graphs.values():
            self.plan.ctx.graph_destroy(g)
        self.graphs = {}

    def ensure_noise(self):
        if self.noise is None:
            p = self.plan
            self.noise = p.alloc(p.rows, self.B * self.C * p.H * p.W, dtype=torch.float32)
        return self.noise

    def _emit_tail(self, stream, with_noise, scale=1.0):
        p = self.plan
        nz = self.noise.data_ptr() if with_noise else None
        # the step kernel also advances the device-side step counter (include/upk.h upk_step_autoadvance)
        p.ctx._chk(p.lib.upk_step_autoadvance(p.hctx, self.step_done.data_ptr()))
        try:
            self._emit_step(stream, nz, scale)
        finally:
            p.ctx._chk(p.lib.upk_step_autoadvance(p.hctx, None))

    def _emit_step(self, stream, nz, scale):
        p = self.plan
        if self.plms:
            require(nz is None, "PLMS runs with eta = 0", ValueError)
            p.ctx._chk(p.lib.upk_plms_step_f32(p.hctx, self.x.data_ptr(), p.eps.data_ptr(), self.coefs.data_ptr(),
                                               p.step.data_ptr(), self.hist.data_ptr(), self.pred_x0.data_ptr(),
                                               p.xin.t.data_ptr(), p.xin.ld, self.B, self.C, p.H * p.W, float(scale),
                                               int(self.cfg), stream))
        elif self.cfg:
            p.ctx._chk(p.lib.upk_ddim_step_cfg_f32(p.hctx, self.x.data_ptr(), p.eps.data_ptr(), self.coefs.data_ptr(), nz,
                                                   p.step.data_ptr(), self.pred_x0.data_ptr(), p.xin.t.data_ptr(),
                                                   p.xin.ld, self.B, self.C, p.H * p.W, float(scale), stream))
        else:
            p.ctx._chk(p.lib.upk_ddim_step_f32(p.hctx, self.x.data_ptr(), p.eps.data_ptr(), self.coefs.data_ptr(), nz,
                                               p.step.data_ptr(), self.pred_x0.data_ptr(), p.xin.t.data_ptr(), p.xin.ld,
                                               p.B, self.C, p.H * p.W, stream))

    def step_eager(self, with_noise, scale=1.0):
        s = self.plan.ctx._s()
        self.plan.body.run(s)
        self._emit_tail(s, with_noise, scale)

    def graph(self, with_noise, scale=1.0, nsteps=1):
        """`nsteps` consecutive DDIM steps captured as ONE HIP graph (static shapes, device-side step index: the same
        graph serves every position of the loop; the guidance scale is a kernel argument, so each scale value gets its
        own graph).  nsteps > 1 saves the graph-to-graph launch gap of the steps inside."""
        key = (with_noise, float(scale) if self.cfg else 1.0) + ((int(nsteps),) if nsteps != 1 else ())
        g = self.graphs.pop(key, None)
        if g is not None:
            self.graphs[key] = g  # (most recently used last: eviction takes the least recently used graph)
        if g is None:
            p = self.plan
            if with_noise:
                self.ensure_noise()
            with L.host_io():  # (no other lane uploads from the host while this thread captures)
                side = torch.cuda.Stream(device=p.dev)
                side.wait_stream(torch.cuda.current_stream(p.dev))
                sp = side.cuda_stream
                p.ctx._chk(p.lib.upk_graph_begin(p.hctx, sp))
                try:
                    for _ in range(int(nsteps)):
                        p.body.run(sp)
                        self._emit_tail(sp, with_noise, scale)
                finally:
                    gh = C.c_void_p()
                    rc = p.lib.upk_graph_end(p.hctx, sp, C.byref(gh))
                p.ctx._chk(rc)
                torch.cuda.current_stream(p.dev).wait_stream(side)
            # one sample() makes up to three graphs per (noise, scale) (full groups of steps, the remainder, single steps
            # around a callback): 16 graphs = five guidance scales in rotation; least recently used first
            if len(self.graphs) >= 16:
                torch.cuda.synchronize(p.dev)  # (the evicted graph may still be executing)
                p.ctx.graph_destroy(self.graphs.pop(next(iter(self.graphs))))
            g = self.graphs[key] = gh
        return g

    def launch(self, with_noise, scale=1.0, nsteps=1):
        p = self.plan
        p.ctx._chk(p.lib.upk_graph_launch(p.hctx, self.graph(with_noise, scale, nsteps), p.ctx._s()))


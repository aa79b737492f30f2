"""fp32 master weights (reference key names) -> operands of the kernels, packed once per weight set:
the implicit-GEMM tile layout of upk_pack_weight_f16 (Packer.pack), and the PackedUNet / PackedVAE* containers the plans lower against."""
import torch

from . import knobs as K
from ._check import require
from .arch import UNetArch, VAEArch


def _rup(v, m):
    return (v + m - 1) // m * m



def head_pad(d):
    for p in (32, 64, 128, 256, 512):
        if d <= p:
            return p
    raise NotImplementedError("attention head dim %d > 512" % d)


class PW:
    """A packed weight: fp16 tiles + fp32 bias in packed row order."""
    __slots__ = ("w", "n_pad", "bias", "n_out", "k_packed", "ksize", "n_real", "k_real", "ln_colsum", "k_append", "w_phase")


class SharedPacks:
    """Packed operands of ONE weight set, built once and shared by the plans of every execution lane (the packs are
    read-only device tensors: only activation buffers, descriptors and graphs are per lane).  wrap(packer) returns a packer
    whose pack() / vec() results are memoised here; `fresh` counts the packs built since the last take_fresh() so that the
    caller can drain the packing stream before another lane's stream reads them."""

    def __init__(self):
        self.d, self.fresh = {}, 0

    def wrap(self, pk):
        return _MemoPacker(pk, self)

    def take_fresh(self):
        n, self.fresh = self.fresh, 0
        return n


class _MemoPacker:
    def __init__(self, pk, shared):
        self._pk, self._sh = pk, shared

    @staticmethod
    def _key(kind, names, kw):
        return (kind, tuple(names) if isinstance(names, (list, tuple)) else names, tuple(sorted(kw.items())))

    def pack(self, names, **kw):
        k = self._key("pack", names, kw)
        if k not in self._sh.d:
            self._sh.d[k] = self._pk.pack(names, **kw)
            self._sh.fresh += 1
        return self._sh.d[k]

    def vec(self, name):
        k = self._key("vec", name, {})
        if k not in self._sh.d:
            self._sh.d[k] = self._pk.vec(name)
            self._sh.fresh += 1
        return self._sh.d[k]

    def __getattr__(self, a):  # (anything else goes to the plain packer, unshared)
        return getattr(self._pk, a)


class Packer:
    """fp32 OIHW / [out,in] master weights -> libupk packed fp16 (done once per weight set)."""

    def __init__(self, ctx, get):
        self.ctx, self.get, self.dev = ctx, get, ctx.device

    def _maps(self, m):
        return None if m is None else torch.as_tensor(m, dtype=torch.int32, device=self.dev).contiguous()

    def pack(self, names, row_map=None, col_map=None, cin_packed=None, bias=True, n_out=None, ln=None):
        """`names`: one weight name or a list whose rows are concatenated (fused q|k|v).
        `ln`: name of a LayerNorm whose affine is folded into this Linear (include/upk.h ln_colsum):
        W' = W * gamma, bias' = bias + W @ beta, plus the column sums of the fp16-rounded W'."""
        if isinstance(names, str):
            names = [names]
        ws = [self.get(n + ".weight") for n in names]
        w = ws[0] if len(ws) == 1 else torch.cat([x.reshape(x.shape[0], -1) for x in ws], 0).reshape(
            -1, *ws[0].shape[1:])
        w = w.contiguous().float()
        ln_bias = None
        if ln is not None:
            require(w.dim() == 2 and col_map is None, "a LayerNorm can only be folded into a Linear without a column map", ValueError)
            gamma, beta = self.get(ln + ".weight").float(), self.get(ln + ".bias").float()
            ln_bias = w @ beta
            w = (w * gamma[None, :]).contiguous()
        rm, cm = self._maps(row_map), self._maps(col_map)
        p = PW()
        p.w, p.n_pad = self.ctx.pack_weight(w, row_map=rm, col_map=cm, cin_packed=cin_packed)
        p.ksize = w.shape[-1] if w.dim() == 4 else 1
        cin = w.shape[1]
        p.k_packed = (cin_packed if cin_packed is not None else _rup(cin if cm is None else cm.numel(), 32))
        n_rows = w.shape[0] if rm is None else rm.numel()
        p.n_out = n_rows if n_out is None else n_out
        p.n_real = w.shape[0] if rm is None else int((rm >= 0).sum().item())
        p.k_real = (cin if cm is None else int((cm >= 0).sum().item())) * p.ksize * p.ksize
        def rows_packed(vec):  # per-output-row vector -> packed row order, zero padded to n_pad
            out = torch.zeros(p.n_pad, dtype=torch.float32, device=self.dev)
            if rm is None:
                out[: vec.numel()] = vec
            else:
                idx = rm.long()
                out[: idx.numel()] = torch.where(idx >= 0, vec[idx.clamp(min=0)], torch.zeros((), device=self.dev))
            return out

        p.bias = None
        p.ln_colsum = None
        p.k_append = 0
        p.w_phase = None
        b = None
        if bias:
            bs = [self.get(n + ".bias") for n in names]
            b = (bs[0] if len(bs) == 1 else torch.cat(bs, 0)).float()
        if ln_bias is not None:
            b = ln_bias if b is None else b + ln_bias
            p.ln_colsum = rows_packed(w.half().float().sum(dim=1))
        if b is not None:
            p.bias = rows_packed(b)
        return p

    def add_upsample_phases(self, p, name):
        """Phase weights of an Upsample conv (include/upk.h w_phase): nearest 2x + conv3x3 = four 2x2 convs on the
        low-resolution grid; tap (ty, tx) of phase (py, px) = sum of the 3x3 taps that read the same low-resolution
        pixel (summed in fp32, rounded to fp16 once)."""
        w = self.get(name + ".weight").float().to(self.dev)
        require(w.dim() == 4 and w.shape[-1] == 3 and w.shape[-2] == 3, "upsample phase weights need a 3x3 conv weight", ValueError)
        taps = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}  # (phase bit, tap) -> 3x3 taps
        parts = []
        for py in (0, 1):
            for px in (0, 1):
                wp = torch.zeros(w.shape[0], w.shape[1], 2, 2, device=self.dev)
                for ty in (0, 1):
                    for tx in (0, 1):
                        for ky in taps[(py, ty)]:
                            for kx in taps[(px, tx)]:
                                wp[:, :, ty, tx] += w[:, :, ky, kx]
                packed, n_pad = self.ctx.pack_weight(wp.contiguous())
                require(n_pad == p.n_pad, "phase weight rows differ from the 3x3 weight's", RuntimeError)
                parts.append(packed.reshape(-1))
        p.w_phase = torch.cat(parts).contiguous()
        return p

    def append_1x1(self, main, skip):
        """`main` followed along K by the 1x1 weight `skip` (include/upk.h: appended K segment): one launch computes
        conv(main) + conv1x1(skip) — a ResBlock's second conv plus its skip projection (openaimodel.py:274-275)."""
        require(skip.ksize == 1 and skip.n_pad == main.n_pad and skip.n_out == main.n_out and main.ln_colsum is None, "append_1x1: the appended weight must be a 1x1 with the main weight's rows", ValueError)
        p = PW()
        p.w = torch.cat([main.w.reshape(-1), skip.w.reshape(-1)])
        p.n_pad, p.n_out, p.ksize, p.k_packed, p.n_real = main.n_pad, main.n_out, main.ksize, main.k_packed, main.n_real
        p.k_append = skip.k_packed
        p.k_real = main.k_real + skip.k_real
        p.ln_colsum = None
        bs = [b for b in (main.bias, skip.bias) if b is not None]
        p.bias = None if not bs else (bs[0] if len(bs) == 1 else bs[0] + bs[1])
        return p

    def pack_product(self, outer, inner):
        """The Linear `inner` followed by the Linear / 1x1 conv `outer` with nothing in between, as ONE weight:
        W = W_outer @ W_inner (fp32, then packed fp16), bias = W_outer @ b_inner (`outer`'s own bias is left to the
        caller: Packer.append_1x1 adds it with the appended segment)."""
        wo = self.get(outer + ".weight").float()
        wo = wo.reshape(wo.shape[0], -1)
        wi = self.get(inner + ".weight").float()
        prod = {"p.weight": (wo @ wi).contiguous(), "p.bias": wo @ self.get(inner + ".bias").float()}
        return Packer(self.ctx, lambda n: prod[n]).pack("p")

    def vec(self, name):
        return self.get(name).float().contiguous()


def qproj_pack(w, gamma, beta, heads, dh, dp, cq, dev):
    """to_q weight [heads*dh, C] (+ the LayerNorm affine in front of it) -> operands of upk_attention_qproj_f16
    (include/upk.h): fp16 [heads][dp (row-permuted)][cq] with gamma folded in, fp32 column sums of the rounded rows and
    W beta, both [heads*dp] in natural order (head dim zero-padded dh -> dp, channels C -> cq)."""
    C_ = w.shape[1]
    wf = torch.zeros(heads, dp, cq, dtype=torch.float32, device=dev)
    wf[:, :dh, :C_] = (w.float().to(dev) * gamma.float().to(dev)[None, :]).reshape(heads, dh, C_)
    bias = torch.zeros(heads, dp, dtype=torch.float32, device=dev)
    bias[:, :dh] = (w.float().to(dev) @ beta.float().to(dev)).reshape(heads, dh)
    w16 = wf.half()
    colsum = w16.float().sum(dim=2)
    r = torch.arange(dp, device=dev)
    kd, t, m = r // 32, (r % 32) // 16, r % 16
    src = 32 * kd + 8 * (m // 4) + 4 * t + (m % 4)  # packed row r holds natural row src
    return w16[:, src, :].contiguous(), colsum.reshape(-1).contiguous(), bias.reshape(-1).contiguous()


def pad_rows_map(parts, heads, dh, dp):
    """Row map of `parts` stacked [heads*dh]-row matrices -> [heads*dp]-row blocks each
    (head dim zero-padded dh -> dp)."""
    j = torch.arange(parts * heads * dp)
    part, r = j // (heads * dp), j % (heads * dp)
    h, d = r // dp, r % dp
    return torch.where(d < dh, part * heads * dh + h * dh + d, torch.full_like(j, -1))


def geglu_rows_map(inner):
    """Per 64 packed rows: [32 value rows | 32 gate rows] (include/upk.h UPK_F_GEGLU)."""
    j = torch.arange(2 * inner)
    blk, r = j // 64, j % 64
    return torch.where(r < 32, blk * 32 + r, inner + blk * 32 + (r - 32))


# ====================================================================== UNet
class PackedUNet:
    """All UNet weights packed for the kernels (independent of batch / resolution)."""

    def __init__(self, ctx, arch: UNetArch, get):
        pk = Packer(ctx, get)
        self.arch = arch
        mc, te = arch.model_channels, arch.time_embed_dim
        if mc % 32:
            raise NotImplementedError("model_channels must be a multiple of 32 (got %d)" % mc)
        w = {}
        w["time_embed.0"] = pk.pack("time_embed.0")
        w["time_embed.2"] = pk.pack("time_embed.2")
        v = {}

        def norm(name):
            v[name] = (pk.vec(name + ".weight"), pk.vec(name + ".bias"))

        for Lr in arch.all_layers():
            n = Lr.name
            if Lr.kind == "conv":
                w[n] = pk.pack(n, cin_packed=_rup(Lr.cin, 32))
            elif Lr.kind == "res":
                norm(n + ".in_layers.0")
                w[n + ".in_layers.2"] = pk.pack(n + ".in_layers.2")
                w[n + ".emb_layers.1"] = pk.pack(n + ".emb_layers.1")
                norm(n + ".out_layers.0")
                w[n + ".out_layers.3"] = pk.pack(n + ".out_layers.3")
                if Lr.cin != Lr.cout:
                    w[n + ".skip_connection"] = pk.pack(n + ".skip_connection")
                    w[n + ".out_layers.3+skip"] = pk.append_1x1(w[n + ".out_layers.3"], w[n + ".skip_connection"])
            elif Lr.kind == "st":
                if Lr.depth != 1:
                    raise NotImplementedError("transformer_depth != 1")
                heads, dh = Lr.heads, Lr.dhead
                dp = head_pad(dh)
                hd = heads * dp
                norm(n + ".norm")
                w[n + ".proj_in"] = pk.pack(n + ".proj_in")
                t = n + ".transformer_blocks.0"
                to_out_cols = pad_rows_map(1, heads, dh, dp)
                # norm1 / norm2 / norm3 can be folded into their only consumers (Emitter.ln_linear decides
                # per shape): both packings are kept, "<name>_ln" has the LayerNorm affine folded in
                for sfx, fold in (("", None), ("_ln", True)):
                    w[t + ".attn1.qkv" + sfx] = pk.pack(
                        [t + ".attn1.to_q", t + ".attn1.to_k", t + ".attn1.to_v"], row_map=pad_rows_map(3, heads, dh, dp),
                        bias=False, n_out=2 * hd, ln=(t + ".norm1") if fold else None)
                    w[t + ".attn2.q" + sfx] = pk.pack(t + ".attn2.to_q", row_map=pad_rows_map(1, heads, dh, dp),
                                                      bias=False, ln=(t + ".norm2") if fold else None)
                    w[t + ".ff.geglu" + sfx] = pk.pack(t + ".ff.net.0.proj", row_map=geglu_rows_map(4 * heads * dh),
                                                       n_out=4 * heads * dh, ln=(t + ".norm3") if fold else None)
                if dp in (32, 64) and Lr.ch % 224 == 0:  # operands of the attention that projects its own queries
                    w[t + ".attn2.qproj"] = qproj_pack(get(t + ".attn2.to_q.weight"), get(t + ".norm2.weight"),
                                                       get(t + ".norm2.bias"), heads, dh, dp, Lr.ch, ctx.device)
                w[t + ".attn1.to_out"] = pk.pack(t + ".attn1.to_out.0", col_map=to_out_cols)
                w[t + ".attn2.kv"] = pk.pack([t + ".attn2.to_k", t + ".attn2.to_v"],
                                             row_map=pad_rows_map(2, heads, dh, dp), bias=False, n_out=hd)
                w[t + ".attn2.to_out"] = pk.pack(t + ".attn2.to_out.0", col_map=to_out_cols)
                # epilogue vectors of the fused head (include/upk.h upk_hblock_desc.vec)
                pi, qkv = w[n + ".proj_in"], w[t + ".attn1.qkv_ln"]
                if pi.n_pad == Lr.ch and qkv.n_pad == 3 * hd and pi.ksize == 1:
                    vec = torch.cat([pi.bias, qkv.ln_colsum, qkv.bias])
                    w[t + ".hblock.vec"] = torch.cat([vec, vec.new_zeros(-vec.numel() % 256)]).contiguous()
                # epilogue vectors of the fused cross-attention half (include/upk.h upk_xblock_desc.vec)
                o1, o2, ql = w[t + ".attn1.to_out"], w[t + ".attn2.to_out"], w[t + ".attn2.q_ln"]
                if o1.n_pad == Lr.ch and o2.n_pad == Lr.ch and ql.n_pad == hd:
                    vec = torch.cat([o1.bias, ql.ln_colsum, ql.bias, o2.bias])
                    w[t + ".xblock.vec"] = torch.cat([vec, vec.new_zeros(-vec.numel() % 256)]).contiguous()
                inner = heads * dh
                w[t + ".ff.out"] = pk.pack(t + ".ff.net.2")
                for k in ("norm1", "norm2", "norm3"):
                    norm(t + "." + k)
                w[n + ".proj_out"] = pk.pack(n + ".proj_out")
                # proj_out(t2 + ff.net.2(h)) = (P F2) h + P t2 + (P b2 + bp): the block's last Linear and the
                # transformer's output projection as one GEMM with t2 as an appended K segment (Emitter.fold_ff_out)
                w[n + ".ff.out+proj_out"] = pk.append_1x1(pk.pack_product(n + ".proj_out", t + ".ff.net.2"),
                                                          w[n + ".proj_out"])
            elif Lr.kind == "down":
                w[n + ".op"] = pk.pack(n + ".op")
            elif Lr.kind == "up":
                w[n + ".conv"] = pk.add_upsample_phases(pk.pack(n + ".conv"), n + ".conv")
        norm("out.0")
        w["out.2"] = pk.pack("out.2")
        self.w, self.v = w, v


# ====================================================================== VAE decoder
class PackedVAEDecoder:
    def __init__(self, ctx, arch: VAEArch, get):
        pk = Packer(ctx, get)
        self.arch = arch
        w, v = {}, {}

        def norm(name):
            v[name] = (pk.vec("decoder." + name + ".weight"), pk.vec("decoder." + name + ".bias"))

        w["post_quant_conv"] = pk.pack("post_quant_conv", cin_packed=_rup(arch.embed_dim, 32))
        for Lr in arch.decoder:
            n, d = Lr.name, "decoder." + Lr.name
            if Lr.kind in ("conv", "conv_out"):
                w[n] = pk.pack(d, cin_packed=_rup(Lr.cin, 32))
            elif Lr.kind == "resnet":
                norm(n + ".norm1")
                w[n + ".conv1"] = pk.pack(d + ".conv1")
                norm(n + ".norm2")
                w[n + ".conv2"] = pk.pack(d + ".conv2")
                if Lr.cin != Lr.cout:
                    w[n + ".nin_shortcut"] = pk.pack(d + ".nin_shortcut")
            elif Lr.kind == "attn":
                if Lr.ch not in (32, 64, 128, 256, 512):
                    raise NotImplementedError("VAE AttnBlock width %d" % Lr.ch)
                norm(n + ".norm")
                w[n + ".qkv"] = pk.pack([d + ".q", d + ".k", d + ".v"], n_out=2 * Lr.ch)
                w[n + ".proj_out"] = pk.pack(d + ".proj_out")
            elif Lr.kind == "upconv":
                w[n] = pk.add_upsample_phases(pk.pack(d), d)
            elif Lr.kind == "norm_out":
                norm(n)
        self.w, self.v = w, v


# ====================================================================== VAE encoder
class PackedVAEEncoder:
    def __init__(self, ctx, arch: VAEArch, get):
        pk = Packer(ctx, get)
        self.arch = arch
        w, v = {}, {}

        def norm(name):
            v[name] = (pk.vec("encoder." + name + ".weight"), pk.vec("encoder." + name + ".bias"))

        for Lr in arch.encoder:
            n, d = Lr.name, "encoder." + Lr.name
            if Lr.kind in ("conv", "conv_out"):
                w[n] = pk.pack(d, cin_packed=_rup(Lr.cin, 32))
            elif Lr.kind == "resnet":
                norm(n + ".norm1")
                w[n + ".conv1"] = pk.pack(d + ".conv1")
                norm(n + ".norm2")
                w[n + ".conv2"] = pk.pack(d + ".conv2")
                if Lr.cin != Lr.cout:
                    w[n + ".nin_shortcut"] = pk.pack(d + ".nin_shortcut")
            elif Lr.kind == "attn":
                if Lr.ch not in (32, 64, 128, 256, 512):
                    raise NotImplementedError("VAE AttnBlock width %d" % Lr.ch)
                norm(n + ".norm")
                w[n + ".qkv"] = pk.pack([d + ".q", d + ".k", d + ".v"], n_out=2 * Lr.ch)
                w[n + ".proj_out"] = pk.pack(d + ".proj_out")
            elif Lr.kind == "downconv":
                w[n] = pk.pack(d)
            elif Lr.kind == "norm_out":
                norm(n)
        zc2 = arch.encoder[-1].cout
        w["quant_conv"] = pk.pack("quant_conv", cin_packed=_rup(zc2, 32))
        self.w, self.v = w, v



"""Architecture description (pure Python, no tensors) of the two networks on the hot path.

The reference builds nn.Module trees in its constructors (openaimodel.py:443-692,
model.py:368-568).  Here an architecture is DATA: a list of layer records plus the
state-dict parameter names/shapes they own.  Two consumers:
  * params.ParamTree   — materialises an nn.Module tree whose state_dict() keys equal the
                         reference's (checkpoint drop-in);
  * engine.*           — lowers the records to a flat program of HIP kernel launches.
"""
from collections import OrderedDict


class Layer(dict):
    """kind + name + attributes; attribute access for readability."""
    __getattr__ = dict.__getitem__

    def __init__(self, kind, name, **kw):
        super().__init__(kind=kind, name=name, **kw)


def _list(v):
    if hasattr(v, "tolist") and not isinstance(v, (list, tuple)):
        v = v.tolist()
    return list(v)


class UNetArch:
    """UNetModel(image_size, in_channels, model_channels, out_channels, num_res_blocks,
    attention_resolutions, ..., use_spatial_transformer=True, context_dim, legacy=False)
    — constructor kwargs of openaimodel.py:443-469.  Only the branch the UPGPT configs
    use is supported; anything else raises NotImplementedError (never a silent fallback)."""

    def __init__(self, image_size=None, in_channels=None, model_channels=None, out_channels=None, num_res_blocks=None,
                 attention_resolutions=None, dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2,
                 num_classes=None, use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1, context_dim=None,
                 n_embed=None, legacy=True):
        if not use_spatial_transformer or context_dim is None:
            raise NotImplementedError("upgpt_amd UNetModel: only use_spatial_transformer=True with a context_dim "
                                      "(every UPGPT config) is implemented")
        if dims != 2 or num_classes is not None or use_scale_shift_norm or resblock_updown or n_embed is not None \
                or not conv_resample:
            raise NotImplementedError("upgpt_amd UNetModel: dims!=2 / num_classes / use_scale_shift_norm / "
                                      "resblock_updown / n_embed / conv_resample=False are not on the UPGPT path")
        if isinstance(context_dim, (list, tuple)) or hasattr(context_dim, "__len__"):
            context_dim = _list(context_dim)
            if len(context_dim) != 1:
                raise NotImplementedError("list-valued context_dim with depth > 1")
            context_dim = context_dim[0]
        if num_heads == -1 and num_head_channels == -1:
            raise AssertionError("Either num_heads or num_head_channels has to be set")
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = mc = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = _list(attention_resolutions)
        self.channel_mult = _list(channel_mult)
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.transformer_depth = transformer_depth
        self.context_dim = context_dim
        self.legacy = legacy
        self.use_checkpoint = use_checkpoint  # accepted, irrelevant for inference (util.py:119-128)
        self.dropout = dropout
        self.time_embed_dim = 4 * mc

        def attn(ch):  # openaimodel.py:542-549: head count / width
            if num_head_channels == -1:
                return num_heads, ch // num_heads
            return ch // num_head_channels, num_head_channels

        def st(name, ch):
            h, d = attn(ch)
            if h * d != ch:
                raise NotImplementedError("SpatialTransformer inner_dim %d != channels %d" % (h * d, ch))
            return Layer("st", name, ch=ch, heads=h, dhead=d, depth=transformer_depth, context_dim=context_dim)

        self.input_blocks = [[Layer("conv", "input_blocks.0.0", cin=in_channels, cout=mc)]]
        chans = [mc]
        ch, ds = mc, 1
        for level, mult in enumerate(self.channel_mult):
            for _ in range(num_res_blocks):
                i = len(self.input_blocks)
                blk = [Layer("res", "input_blocks.%d.0" % i, cin=ch, cout=mult * mc)]
                ch = mult * mc
                if ds in self.attention_resolutions:
                    blk.append(st("input_blocks.%d.1" % i, ch))
                self.input_blocks.append(blk)
                chans.append(ch)
            if level != len(self.channel_mult) - 1:
                i = len(self.input_blocks)
                self.input_blocks.append([Layer("down", "input_blocks.%d.0" % i, ch=ch)])
                chans.append(ch)
                ds *= 2
        self.middle_block = [Layer("res", "middle_block.0", cin=ch, cout=ch), st("middle_block.1", ch),
                             Layer("res", "middle_block.2", cin=ch, cout=ch)]
        self.output_blocks = []
        for level, mult in list(enumerate(self.channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                skip = chans.pop()
                o = len(self.output_blocks)
                blk = [Layer("res", "output_blocks.%d.0" % o, cin=ch + skip, cout=mc * mult, split=(ch, skip))]
                ch = mc * mult
                if ds in self.attention_resolutions:
                    blk.append(st("output_blocks.%d.%d" % (o, len(blk)), ch))
                if level and i == num_res_blocks:
                    blk.append(Layer("up", "output_blocks.%d.%d" % (o, len(blk)), ch=ch))
                    ds //= 2
                self.output_blocks.append(blk)
        self.final_channels = ch

    def all_layers(self):
        for blk in self.input_blocks:
            yield from blk
        yield from self.middle_block
        for blk in self.output_blocks:
            yield from blk

    def param_shapes(self):
        """name -> shape, in the reference's registration order."""
        mc, te = self.model_channels, self.time_embed_dim
        p = OrderedDict()

        def wb(name, *wshape):
            p[name + ".weight"] = tuple(wshape)
            p[name + ".bias"] = (wshape[0],)

        def norm(name, c):
            p[name + ".weight"] = (c,)
            p[name + ".bias"] = (c,)

        wb("time_embed.0", te, mc)
        wb("time_embed.2", te, te)
        for L in self.all_layers():
            n = L.name
            if L.kind == "conv":
                wb(n, L.cout, L.cin, 3, 3)
            elif L.kind == "res":
                norm(n + ".in_layers.0", L.cin)
                wb(n + ".in_layers.2", L.cout, L.cin, 3, 3)
                wb(n + ".emb_layers.1", L.cout, te)
                norm(n + ".out_layers.0", L.cout)
                wb(n + ".out_layers.3", L.cout, L.cout, 3, 3)
                if L.cin != L.cout:
                    wb(n + ".skip_connection", L.cout, L.cin, 1, 1)
            elif L.kind == "st":
                c, inner, cd = L.ch, L.heads * L.dhead, L.context_dim
                norm(n + ".norm", c)
                wb(n + ".proj_in", inner, c, 1, 1)
                for d in range(L.depth):
                    t = n + ".transformer_blocks.%d" % d
                    for a, kd in (("attn1", inner), ("attn2", cd)):
                        p[t + "." + a + ".to_q.weight"] = (inner, inner)
                        p[t + "." + a + ".to_k.weight"] = (inner, kd)
                        p[t + "." + a + ".to_v.weight"] = (inner, kd)
                        wb(t + "." + a + ".to_out.0", inner, inner)
                        if a == "attn1":  # registration order: attn1, ff, attn2, norms (attention.py:198-205)
                            wb(t + ".ff.net.0.proj", 8 * inner, inner)
                            wb(t + ".ff.net.2", inner, 4 * inner)
                    for k in ("norm1", "norm2", "norm3"):
                        norm(t + "." + k, inner)
                wb(n + ".proj_out", c, inner, 1, 1)
            elif L.kind == "down":
                wb(n + ".op", L.ch, L.ch, 3, 3)
            elif L.kind == "up":
                wb(n + ".conv", L.ch, L.ch, 3, 3)
        norm("out.0", self.final_channels)
        wb("out.2", self.out_channels, mc, 3, 3)
        return p

    def flops(self, batch, h, w, n_ctx):
        """Algorithmic multiply-add FLOPs (x2) of conv / Linear / QK^T / PV per forward,
        counted the way BASELINE.md §2 does (un-padded dims, biases/norms excluded)."""
        te, mc = self.time_embed_dim, self.model_channels
        f = 2 * batch * (mc * te + te * te)
        hw = {"h": h, "w": w}

        def pix():
            return batch * hw["h"] * hw["w"]

        for L in self.all_layers():
            if L.kind == "conv":
                f += 2 * pix() * L.cin * L.cout * 9
            elif L.kind == "res":
                f += 2 * pix() * 9 * (L.cin * L.cout + L.cout * L.cout) + 2 * batch * te * L.cout
                if L.cin != L.cout:
                    f += 2 * pix() * L.cin * L.cout
            elif L.kind == "st":
                n = hw["h"] * hw["w"]
                c, cd = L.ch, L.context_dim
                f += 2 * pix() * c * c * 2  # proj_in / proj_out
                per = 0
                per += 2 * pix() * c * c * 4                      # attn1 q,k,v,out
                per += 2 * pix() * c * c * 2                      # attn2 q,out
                per += 2 * batch * n_ctx * cd * c * 2              # attn2 k,v
                per += 2 * pix() * c * 8 * c + 2 * pix() * 4 * c * c  # GEGLU + FF out
                per += 2 * 2 * batch * n * n * c                   # self QK^T + PV (all heads)
                per += 2 * 2 * batch * n * n_ctx * c               # cross
                f += per * L.depth
            elif L.kind == "down":
                hw["h"] = (hw["h"] + 2 - 3) // 2 + 1
                hw["w"] = (hw["w"] + 2 - 3) // 2 + 1
                f += 2 * pix() * L.ch * L.ch * 9
            elif L.kind == "up":
                hw["h"] *= 2
                hw["w"] *= 2
                f += 2 * pix() * L.ch * L.ch * 9
        f += 2 * pix() * mc * self.out_channels * 9
        return f


def unet_layer_roofline(arch, batch, h, w, n_ctx, peak_flops=2.5e15, peak_bw=8.0e12):
    """SURVEY.md 8(d)'s honest ceiling of one UNet forward: sum over the GEMM-shaped ops (conv, Linear, QK^T, PV) of
    max(FLOP / dense fp16 MFMA peak, un-fused fp16 bytes / HBM peak) — each op reads its operands and writes its result
    once.  Returns (seconds, flops, bytes).  About 42 % of the FLOPs of bbox.yaml at bs = 8 sit in ops whose weight
    read takes longer than their MFMA time, which is why the MFMA-only figure is not the yardstick."""
    te, mc = arch.time_embed_dim, arch.model_channels
    tot = [0.0, 0, 0]

    def op(flops, nbytes):
        tot[0] += max(flops / peak_flops, nbytes / peak_bw)
        tot[1] += flops
        tot[2] += nbytes

    def gemm(M, K, N):
        op(2 * M * K * N, 2 * (M * K + K * N + M * N))

    def attn(B, nq, nkv, c):  # all heads: QK^T + PV; q, k, v read, out written
        op(2 * 2 * B * nq * nkv * c, 2 * (2 * B * nq * c + 2 * B * nkv * c))

    gemm(batch, mc, te)
    gemm(batch, te, te)
    hw = [h, w]
    pix = lambda: batch * hw[0] * hw[1]
    for L in arch.all_layers():
        if L.kind == "conv":
            gemm(pix(), 9 * L.cin, L.cout)
        elif L.kind == "res":
            gemm(pix(), 9 * L.cin, L.cout)
            gemm(batch, te, L.cout)
            gemm(pix(), 9 * L.cout, L.cout)
            if L.cin != L.cout:
                gemm(pix(), L.cin, L.cout)
        elif L.kind == "st":
            n, c, cd = hw[0] * hw[1], L.ch, L.context_dim
            gemm(pix(), c, c)  # proj_in
            for _ in range(L.depth):
                for _ in range(4):  # attn1 q, k, v, out
                    gemm(pix(), c, c)
                attn(batch, n, n, c)
                gemm(pix(), c, c)  # attn2 q
                gemm(batch * n_ctx, cd, c)
                gemm(batch * n_ctx, cd, c)
                attn(batch, n, n_ctx, c)
                gemm(pix(), c, c)  # attn2 out
                gemm(pix(), c, 8 * c)  # GEGLU projection
                gemm(pix(), 4 * c, c)  # FF out
            gemm(pix(), c, c)  # proj_out
        elif L.kind == "down":
            hw[0], hw[1] = (hw[0] + 2 - 3) // 2 + 1, (hw[1] + 2 - 3) // 2 + 1
            gemm(pix(), 9 * L.ch, L.ch)
        elif L.kind == "up":
            hw[0], hw[1] = hw[0] * 2, hw[1] * 2
            gemm(pix(), 9 * L.ch, L.ch)
    gemm(pix(), 9 * mc, arch.out_channels)
    return tot[0], tot[1], tot[2]


class VAEArch:
    """AutoencoderKL ddconfig (autoencoder.py:286-306, model.py Encoder 368-432 / Decoder 462-533)."""

    def __init__(self, ddconfig, embed_dim):
        dd = dict(ddconfig)
        self.dd = dd
        self.embed_dim = embed_dim
        self.ch = dd["ch"]
        self.ch_mult = _list(dd["ch_mult"])
        self.nrb = dd["num_res_blocks"]
        self.z_channels = dd["z_channels"]
        self.in_channels = dd["in_channels"]
        self.out_ch = dd["out_ch"]
        self.resolution = dd["resolution"]
        self.attn_resolutions = _list(dd.get("attn_resolutions", []))
        self.double_z = dd.get("double_z", True)
        if dd.get("attn_type", "vanilla") != "vanilla" or dd.get("use_linear_attn", False):
            raise NotImplementedError("only vanilla AttnBlock is on the UPGPT path")
        if not dd.get("resamp_with_conv", True):
            raise NotImplementedError("resamp_with_conv=False")
        self.nres = len(self.ch_mult)
        self.factor = 2 ** (self.nres - 1)

        # ---- decoder records (model.py:535-568 order of execution)
        block_in = self.ch * self.ch_mult[-1]
        curr = self.resolution // self.factor
        dec = [Layer("conv", "conv_in", cin=self.z_channels, cout=block_in),
               Layer("resnet", "mid.block_1", cin=block_in, cout=block_in),
               Layer("attn", "mid.attn_1", ch=block_in),
               Layer("resnet", "mid.block_2", cin=block_in, cout=block_in)]
        for lvl in reversed(range(self.nres)):
            block_out = self.ch * self.ch_mult[lvl]
            for ib in range(self.nrb + 1):
                dec.append(Layer("resnet", "up.%d.block.%d" % (lvl, ib), cin=block_in, cout=block_out))
                block_in = block_out
                if curr in self.attn_resolutions:
                    dec.append(Layer("attn", "up.%d.attn.%d" % (lvl, ib), ch=block_in))
            if lvl != 0:
                dec.append(Layer("upconv", "up.%d.upsample.conv" % lvl, ch=block_in))
                curr *= 2
        dec.append(Layer("norm_out", "norm_out", ch=block_in))
        dec.append(Layer("conv_out", "conv_out", cin=block_in, cout=self.out_ch))
        self.decoder = dec

        # ---- encoder records (model.py:434-459)
        enc = [Layer("conv", "conv_in", cin=self.in_channels, cout=self.ch)]
        in_mult = [1] + self.ch_mult
        curr = self.resolution
        block_in = self.ch
        for lvl in range(self.nres):
            block_in = self.ch * in_mult[lvl]
            block_out = self.ch * self.ch_mult[lvl]
            for ib in range(self.nrb):
                enc.append(Layer("resnet", "down.%d.block.%d" % (lvl, ib), cin=block_in, cout=block_out))
                block_in = block_out
                if curr in self.attn_resolutions:
                    enc.append(Layer("attn", "down.%d.attn.%d" % (lvl, ib), ch=block_in))
            if lvl != self.nres - 1:
                enc.append(Layer("downconv", "down.%d.downsample.conv" % lvl, ch=block_in))
                curr //= 2
        enc += [Layer("resnet", "mid.block_1", cin=block_in, cout=block_in), Layer("attn", "mid.attn_1", ch=block_in),
                Layer("resnet", "mid.block_2", cin=block_in, cout=block_in), Layer("norm_out", "norm_out", ch=block_in),
                Layer("conv_out", "conv_out", cin=block_in,
                      cout=2 * self.z_channels if self.double_z else self.z_channels)]
        self.encoder = enc

    @staticmethod
    def _shapes(records, prefix, p):
        def wb(name, *wshape):
            p[prefix + name + ".weight"] = tuple(wshape)
            p[prefix + name + ".bias"] = (wshape[0],)

        def norm(name, c):
            p[prefix + name + ".weight"] = (c,)
            p[prefix + name + ".bias"] = (c,)

        for L in records:
            n = L.name
            if L.kind in ("conv", "conv_out"):
                wb(n, L.cout, L.cin, 3, 3)
            elif L.kind == "resnet":
                norm(n + ".norm1", L.cin)
                wb(n + ".conv1", L.cout, L.cin, 3, 3)
                norm(n + ".norm2", L.cout)
                wb(n + ".conv2", L.cout, L.cout, 3, 3)
                if L.cin != L.cout:
                    wb(n + ".nin_shortcut", L.cout, L.cin, 1, 1)
            elif L.kind == "attn":
                norm(n + ".norm", L.ch)
                for k in ("q", "k", "v", "proj_out"):
                    wb(n + "." + k, L.ch, L.ch, 1, 1)
            elif L.kind in ("upconv", "downconv"):
                wb(n, L.ch, L.ch, 3, 3)
            elif L.kind == "norm_out":
                norm(n, L.ch)

    def param_shapes(self):
        p = OrderedDict()
        self._shapes(self.encoder, "encoder.", p)
        self._shapes(self.decoder, "decoder.", p)
        zc = self.z_channels
        p["quant_conv.weight"] = (2 * self.embed_dim, 2 * zc, 1, 1)
        p["quant_conv.bias"] = (2 * self.embed_dim,)
        p["post_quant_conv.weight"] = (zc, self.embed_dim, 1, 1)
        p["post_quant_conv.bias"] = (zc,)
        return p

    def decoder_flops(self, batch, h, w):
        f = 2 * batch * h * w * self.embed_dim * self.z_channels
        for L in self.decoder:
            pix = batch * h * w
            if L.kind in ("conv", "conv_out"):
                f += 2 * pix * 9 * L.cin * L.cout
            elif L.kind == "resnet":
                f += 2 * pix * 9 * (L.cin * L.cout + L.cout * L.cout)
                if L.cin != L.cout:
                    f += 2 * pix * L.cin * L.cout
            elif L.kind == "attn":
                f += 2 * pix * L.ch * L.ch * 4 + 2 * 2 * batch * (h * w) * (h * w) * L.ch
            elif L.kind == "upconv":
                h, w = 2 * h, 2 * w
                f += 2 * batch * h * w * 9 * L.ch * L.ch
        return f

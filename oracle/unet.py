"""CPU fp32 restatement of the reference denoiser (test oracle, functional style).

unet_forward(sd, cfg, x, t, context) evaluates the reference UNetModel on a plain
state-dict `sd` (reference key names under `prefix`), following
  ldm/modules/diffusionmodules/openaimodel.py  UNetModel.__init__ 443-692 / forward 710-742,
      ResBlock._forward 255-275, Downsample 134-160, Upsample 91-119,
  ldm/modules/attention.py  SpatialTransformer 218-261, BasicTransformerBlock 196-215,
      CrossAttention 152-193, GEGLU/FeedForward 37-64,
  ldm/modules/diffusionmodules/util.py  timestep_embedding 151-171, GroupNorm32 214-216.
Only the branch every UPGPT config uses is restated (use_spatial_transformer=True,
legacy=False, resblock_updown=False, use_scale_shift_norm=False, num_classes=None, dims=2).
"""
import math

import torch
import torch.nn.functional as F


def timestep_embedding(t, dim, max_period=10000):
    """util.py:151-171 — [cos | sin] order, fp32 frequencies."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def unet_layout(cfg):
    """Replays the constructor loops (openaimodel.py:513-680) and returns, per block,
    the list of (kind, name, params) layers.  kinds: conv, res, st, down, up."""
    mc = cfg["model_channels"]
    mult = list(cfg.get("channel_mult", (1, 2, 4, 8)))
    nrb = cfg["num_res_blocks"]
    attn_res = list(cfg["attention_resolutions"])
    heads = cfg.get("num_heads", -1)
    nhc = cfg.get("num_head_channels", -1)

    def st_dims(ch):  # openaimodel.py:542-549 with legacy=False
        if nhc == -1:
            return heads, ch // heads
        return ch // nhc, nhc

    inputs = [[("conv", "input_blocks.0.0", dict(cin=cfg["in_channels"], cout=mc))]]
    chans = [mc]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            i = len(inputs)
            layers = [("res", "input_blocks.%d.0" % i, dict(cin=ch, cout=m * mc))]
            ch = m * mc
            if ds in attn_res:
                h, dh = st_dims(ch)
                layers.append(("st", "input_blocks.%d.1" % i, dict(ch=ch, heads=h, dhead=dh)))
            inputs.append(layers)
            chans.append(ch)
        if level != len(mult) - 1:
            i = len(inputs)
            inputs.append([("down", "input_blocks.%d.0" % i, dict(ch=ch))])
            chans.append(ch)
            ds *= 2
    h, dh = st_dims(ch)
    middle = [("res", "middle_block.0", dict(cin=ch, cout=ch)),
              ("st", "middle_block.1", dict(ch=ch, heads=h, dhead=dh)),
              ("res", "middle_block.2", dict(cin=ch, cout=ch))]
    outputs = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            o = len(outputs)
            layers = [("res", "output_blocks.%d.0" % o, dict(cin=ch + ich, cout=mc * m))]
            ch = mc * m
            if ds in attn_res:
                h, dh = st_dims(ch)
                layers.append(("st", "output_blocks.%d.%d" % (o, len(layers)), dict(ch=ch, heads=h, dhead=dh)))
            if level and i == nrb:
                layers.append(("up", "output_blocks.%d.%d" % (o, len(layers)), dict(ch=ch)))
                ds //= 2
            outputs.append(layers)
    return inputs, middle, outputs


def _gn(sd, name, x, eps):
    return F.group_norm(x.float(), 32, sd[name + ".weight"], sd[name + ".bias"], eps).type(x.dtype)


def _conv(sd, name, x, stride=1, padding=1):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=padding)


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def resblock(sd, p, x, emb):
    """openaimodel.py:255-275 (non-updown, no scale-shift): GN(1e-5)+SiLU+conv3x3,
    + Linear(SiLU(emb)) broadcast, GN+SiLU+conv3x3, + skip (identity | conv1x1)."""
    h = _conv(sd, p + ".in_layers.2", F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5)))
    e = _lin(sd, p + ".emb_layers.1", F.silu(emb))
    h = h + e[:, :, None, None]
    h = _conv(sd, p + ".out_layers.3", F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5)))
    if (p + ".skip_connection.weight") in sd:
        x = _conv(sd, p + ".skip_connection", x, padding=0)
    return x + h


def cross_attention(sd, p, x, context, heads):
    """attention.py:170-193: q/k/v without bias, scale d_head^-0.5, softmax over keys,
    to_out Linear with bias."""
    ctx = x if context is None else context
    q, k, v = _lin(sd, p + ".to_q", x), _lin(sd, p + ".to_k", ctx), _lin(sd, p + ".to_v", ctx)
    b, n, inner = q.shape
    d = inner // heads
    split = lambda t: t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], d)
    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bid,bjd->bij", q, k) * (d ** -0.5)
    out = torch.einsum("bij,bjd->bid", sim.softmax(dim=-1), v)
    out = out.reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, inner)
    return _lin(sd, p + ".to_out.0", out)


def transformer_block(sd, p, x, context, heads):
    """attention.py:211-215 + GEGLU feed-forward 42-44, 63-64 (LayerNorm eps 1e-5)."""
    ln = lambda nm, t: F.layer_norm(t, (t.shape[-1],), sd[p + nm + ".weight"], sd[p + nm + ".bias"], 1e-5)
    x = cross_attention(sd, p + ".attn1", ln(".norm1", x), None, heads) + x
    x = cross_attention(sd, p + ".attn2", ln(".norm2", x), context, heads) + x
    hcat = _lin(sd, p + ".ff.net.0.proj", ln(".norm3", x))
    val, gate = hcat.chunk(2, dim=-1)
    return _lin(sd, p + ".ff.net.2", val * F.gelu(gate)) + x


def spatial_transformer(sd, p, x, context, heads, depth=1):
    """attention.py:250-261: GN(eps 1e-6) -> 1x1 -> tokens -> blocks -> 1x1 -> + input."""
    b, c, h, w = x.shape
    t = _conv(sd, p + ".proj_in", _gn(sd, p + ".norm", x, 1e-6), padding=0)
    t = t.permute(0, 2, 3, 1).reshape(b, h * w, t.shape[1])
    for d in range(depth):
        t = transformer_block(sd, p + ".transformer_blocks.%d" % d, t, context, heads)
    t = t.reshape(b, h, w, -1).permute(0, 3, 1, 2)
    return _conv(sd, p + ".proj_out", t, padding=0) + x


def _run_layers(sd, prefix, layers, h, emb, context, depth):
    for kind, name, p in layers:
        n = prefix + name
        if kind == "conv":
            h = _conv(sd, n, h)
        elif kind == "res":
            h = resblock(sd, n, h, emb)
        elif kind == "st":
            h = spatial_transformer(sd, n, h, context, p["heads"], depth)
        elif kind == "down":  # openaimodel.py:150-160 conv3x3 stride 2 pad 1
            h = _conv(sd, n + ".op", h, stride=2)
        elif kind == "up":  # openaimodel.py:109-119 nearest x2 then conv3x3
            h = _conv(sd, n + ".conv", F.interpolate(h, scale_factor=2, mode="nearest"))
    return h


@torch.no_grad()
def unet_forward(sd, cfg, x, timesteps, context, prefix="model.diffusion_model.", taps=None):
    """openaimodel.py:710-742. `taps` (dict) receives per-block outputs for debugging."""
    inputs, middle, outputs = unet_layout(cfg)
    depth = cfg.get("transformer_depth", 1)
    temb = timestep_embedding(timesteps, cfg["model_channels"])
    emb = _lin(sd, prefix + "time_embed.2", F.silu(_lin(sd, prefix + "time_embed.0", temb)))
    hs = []
    h = x.float()
    for i, layers in enumerate(inputs):
        h = _run_layers(sd, prefix, layers, h, emb, context, depth)
        hs.append(h)
        if taps is not None:
            taps["input_blocks.%d" % i] = h
    h = _run_layers(sd, prefix, middle, h, emb, context, depth)
    if taps is not None:
        taps["middle_block"] = h
    for i, layers in enumerate(outputs):
        h = torch.cat([h, hs.pop()], dim=1)  # current features first (openaimodel.py:736)
        h = _run_layers(sd, prefix, layers, h, emb, context, depth)
        if taps is not None:
            taps["output_blocks.%d" % i] = h
    h = F.silu(_gn(sd, prefix + "out.0", h, 1e-5))
    return _conv(sd, prefix + "out.2", h)


@torch.no_grad()
def diffusion_wrapper(sd, cfg, x, t, c_concat=None, c_crossattn=None, conditioning_key="hybrid",
                      prefix="model.diffusion_model."):
    """ddpm.py:1557-1577 DiffusionWrapper.forward."""
    if conditioning_key is None:
        return unet_forward(sd, cfg, x, t, None, prefix)
    if conditioning_key == "concat":
        return unet_forward(sd, cfg, torch.cat([x] + c_concat, dim=1), t, None, prefix)
    if conditioning_key == "crossattn":
        return unet_forward(sd, cfg, x, t, torch.cat(c_crossattn, 1), prefix)
    if conditioning_key == "hybrid":  # c_crossattn is a TENSOR here (ddpm.py:1569)
        return unet_forward(sd, cfg, torch.cat([x] + c_concat, dim=1), t, torch.cat([c_crossattn], 1), prefix)
    raise NotImplementedError(conditioning_key)

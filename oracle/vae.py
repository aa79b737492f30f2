"""CPU fp32 restatement of the first-stage decode (test oracle).

Follows ldm/models/autoencoder.py:330-333 (AutoencoderKL.decode = post_quant_conv -> Decoder)
and ldm/modules/diffusionmodules/model.py: Decoder 462-568, ResnetBlock 82-141,
AttnBlock 150-202, Upsample 42-57, Normalize 38-39 (GroupNorm 32 groups, eps 1e-6),
nonlinearity 33-35 (swish).
"""
import torch
import torch.nn.functional as F


def _gn(sd, name, x):
    return F.group_norm(x, 32, sd[name + ".weight"], sd[name + ".bias"], 1e-6)


def _conv(sd, name, x, padding=1):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=padding)


def resnet_block(sd, p, x):
    """model.py:116-141 with temb=None."""
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x)))
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h)))
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x, padding=0)
    elif (p + ".conv_shortcut.weight") in sd:
        x = _conv(sd, p + ".conv_shortcut", x)
    return x + h


def attn_block(sd, p, x):
    """model.py:177-202: single-head attention over pixels, scale c^-0.5."""
    h = _gn(sd, p + ".norm", x)
    q, k, v = (_conv(sd, p + n, h, padding=0) for n in (".q", ".k", ".v"))
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w = torch.bmm(q, k) * (int(c) ** -0.5)
    w = F.softmax(w, dim=2)
    v = v.reshape(b, c, hh * ww)
    o = torch.bmm(v, w.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, p + ".proj_out", o, padding=0)


@torch.no_grad()
def decoder_forward(sd, dd, z, prefix="first_stage_model.decoder."):
    """model.py:535-568."""
    nres = len(dd["ch_mult"])
    nrb = dd["num_res_blocks"]
    attn_resolutions = list(dd.get("attn_resolutions", []))
    curr_res = dd["resolution"] // 2 ** (nres - 1)
    h = _conv(sd, prefix + "conv_in", z)
    h = resnet_block(sd, prefix + "mid.block_1", h)
    h = attn_block(sd, prefix + "mid.attn_1", h)
    h = resnet_block(sd, prefix + "mid.block_2", h)
    for lvl in reversed(range(nres)):
        for ib in range(nrb + 1):
            h = resnet_block(sd, prefix + "up.%d.block.%d" % (lvl, ib), h)
            if curr_res in attn_resolutions:
                h = attn_block(sd, prefix + "up.%d.attn.%d" % (lvl, ib), h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, prefix + "up.%d.upsample.conv" % lvl, h)
            curr_res *= 2
    h = F.silu(_gn(sd, prefix + "norm_out", h))
    return _conv(sd, prefix + "conv_out", h)


@torch.no_grad()
def decode_first_stage(sd, dd, z, scale_factor=0.18215, prefix="first_stage_model."):
    """ddpm.py:779 (z / scale_factor) + autoencoder.py:330-333."""
    z = 1.0 / scale_factor * z
    z = _conv(sd, prefix + "post_quant_conv", z, padding=0)
    return decoder_forward(sd, dd, z, prefix + "decoder.")


@torch.no_grad()
def encoder_forward(sd, dd, x, prefix="first_stage_model.encoder."):
    """model.py:434-459 (Encoder.forward, temb=None).  Downsample = pad (0,1,0,1) then
    conv3x3 stride 2 without padding (model.py:70-76)."""
    nres = len(dd["ch_mult"])
    nrb = dd["num_res_blocks"]
    attn_resolutions = list(dd.get("attn_resolutions", []))
    curr_res = dd["resolution"]
    h = _conv(sd, prefix + "conv_in", x)
    for lvl in range(nres):
        for ib in range(nrb):
            h = resnet_block(sd, prefix + "down.%d.block.%d" % (lvl, ib), h)
            if curr_res in attn_resolutions:
                h = attn_block(sd, prefix + "down.%d.attn.%d" % (lvl, ib), h)
        if lvl != nres - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            n = prefix + "down.%d.downsample.conv" % lvl
            h = F.conv2d(h, sd[n + ".weight"], sd[n + ".bias"], stride=2, padding=0)
            curr_res //= 2
    h = resnet_block(sd, prefix + "mid.block_1", h)
    h = attn_block(sd, prefix + "mid.attn_1", h)
    h = resnet_block(sd, prefix + "mid.block_2", h)
    h = F.silu(_gn(sd, prefix + "norm_out", h))
    return _conv(sd, prefix + "conv_out", h)


@torch.no_grad()
def encode_moments(sd, dd, x, prefix="first_stage_model."):
    """autoencoder.py:324-328: moments = quant_conv(encoder(x)); the posterior is
    DiagonalGaussianDistribution(moments) (distributions.py:24-36: mean | logvar clamp [-30, 20])."""
    h = encoder_forward(sd, dd, x, prefix + "encoder.")
    return _conv(sd, prefix + "quant_conv", h, padding=0)

"""Synthetic weights and inputs by RECIPE (there are no checkpoints / datasets offline).

Weights are a pure function of the state-dict key name and shape, so the reference
(imported in the build container by tests/golden/make_goldens.py), the CPU oracle and
the HIP path all see bit-identical fp32 tensors without shipping 1.7 GB fixtures
(SURVEY.md §8c-1).  Zero-initialised modules of the reference (openaimodel.py:229-231,
685; attention.py:244-248) are re-randomised too, otherwise a random-init UNet outputs
exactly 0 and parity tests would test nothing.

Inputs follow the DeepFashion value distributions of SURVEY.md §8d.
"""
import math
import zlib

import numpy as np
import torch

RECIPE_VERSION = 1


def _gen(key, salt=0):
    return torch.Generator(device="cpu").manual_seed((zlib.crc32(key.encode()) + 7919 * salt) & 0x7FFFFFFF)


def synth_tensor(key, shape, salt=0):
    """fp32 CPU tensor for state-dict entry `key`."""
    shape = tuple(shape)
    g = _gen(key, salt)
    r = torch.randn(shape, generator=g, dtype=torch.float32)
    if len(shape) >= 2:  # conv OIHW / linear [out, in]
        fan_in = int(np.prod(shape[1:]))
        return r * (1.0 / math.sqrt(fan_in))
    if key.endswith("weight"):  # GroupNorm / LayerNorm gain
        return 1.0 + 0.1 * r
    return 0.02 * r  # biases


def synth_state_dict(shapes, prefixes=("model.diffusion_model.", "first_stage_model.", "extra_cond_models."),
                     salt=0):
    """{key: tensor} for every key in `shapes` (dict key -> shape) under `prefixes`."""
    return {k: synth_tensor(k, s, salt) for k, s in sorted(shapes.items())
            if any(k.startswith(p) for p in prefixes)}


def fill_module_(module, prefixes=("model.diffusion_model.", "first_stage_model.", "extra_cond_models."), salt=0,
                 ema=True):
    """Loads recipe weights into an nn.Module whose state-dict uses the reference key names.
    With ema=True the LitEma shadow buffers (model_ema.<name without dots>, ema.py:15-20)
    receive the SAME tensors as the live weights, so ema_scope() is numerically a no-op."""
    sd = module.state_dict()
    new = synth_state_dict({k: tuple(v.shape) for k, v in sd.items()}, prefixes, salt)
    if ema:
        for k, v in list(new.items()):
            if k.startswith("model."):
                s_name = "model_ema." + k[len("model."):].replace(".", "")
                if s_name in sd:
                    new[s_name] = v.clone()
    missing, unexpected = module.load_state_dict(new, strict=False)
    if unexpected:  # (plain raise: this module is also loaded stand-alone by tests/golden/make_goldens.py)
        raise RuntimeError("unexpected keys: %s" % (unexpected,))
    return new


def fill_ema_(module, salt=1):
    """Gives the LitEma shadow buffers their OWN recipe draw (salt != 0: different from the live weights), so that a test
    can tell whether ema_scope() was honoured.  Key = the live parameter's name, as in fill_module_."""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        if k.startswith("model.diffusion_model."):
            s_name = "model_ema." + k[len("model."):].replace(".", "")
            if s_name in sd:
                new[s_name] = synth_tensor(k, tuple(v.shape), salt)
    module.load_state_dict(new, strict=False)
    return new


def crc_of(t):
    return zlib.crc32(t.detach().cpu().contiguous().numpy().tobytes()) & 0xFFFFFFFF


def person_mask(batch, h, w):
    """Bug-compatible bbox mask values (deepfashion_inshop.py:235-239): -1 outside,
    -0.99215686 inside a centred box rows h/8..7h/8, cols w/4..3w/4."""
    m = torch.full((batch, 1, h, w), -1.0)
    m[:, :, h // 8: 7 * h // 8, w // 4: 3 * w // 4] = -0.99215686
    return m


def synth_inputs(batch, latent_hw=(32, 24), channels=4, ctx_tokens=87, ctx_dim=768, seed=0, text_only=False,
                 concat_channels=1, steps=0):
    """x_T, c_crossattn [B, 87, 768] (text | style | pose), c_concat [B, 1, h, w] and
    optional per-step noise [steps, B, C, h, w] from one CPU generator."""
    g = torch.Generator(device="cpu").manual_seed(1234 + seed)
    h, w = latent_hw
    x_T = torch.randn(batch, channels, h, w, generator=g)
    n_txt = min(77, ctx_tokens)
    txt = torch.randn(batch, n_txt, ctx_dim, generator=g)
    n_style = max(0, min(9, ctx_tokens - n_txt))
    n_pose = max(0, ctx_tokens - n_txt - n_style)
    style = 0.45 * torch.randn(batch, n_style, ctx_dim, generator=g)
    pose = 0.5 * torch.randn(batch, n_pose, ctx_dim, generator=g)
    if text_only:  # "null style": one constant vector repeated, zero SMPL
        style = style[:1, :1].expand(batch, n_style, ctx_dim).clone()
        pose = torch.zeros_like(pose)
    ctx = torch.cat([txt, style, pose], dim=1)
    if concat_channels == 1:
        cc = person_mask(batch, h, w)
    else:  # upscale model: low-res image resized to the latent size, U(-1, 1)
        cc = torch.rand(batch, concat_channels, h, w, generator=g) * 2 - 1
    noise = torch.randn(steps, batch, channels, h, w, generator=g) if steps else None
    return {"x_T": x_T, "c_crossattn": ctx, "c_concat": cc, "noise": noise}


# The reference model configs, restated as plain dicts (configs/deepfashion/bbox.yaml:45-79,
# models/upgpt/upscale/config.yaml:37-76) — used where a YAML parser/config file is not wanted.
BBOX_UNET = dict(image_size=32, in_channels=5, out_channels=4, model_channels=224, attention_resolutions=[4, 2, 1],
                 num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                 transformer_depth=1, context_dim=768, use_checkpoint=True, legacy=False)
BBOX_DDCONFIG = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                     ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
UPSCALE_UNET = dict(image_size=32, in_channels=6, out_channels=3, model_channels=256, attention_resolutions=[2, 4, 8],
                    num_res_blocks=2, channel_mult=[1, 2, 2, 4], num_heads=8, use_spatial_transformer=True,
                    transformer_depth=1, context_dim=768, use_checkpoint=True, legacy=False)
UPSCALE_DDCONFIG = dict(double_z=True, z_channels=3, resolution=256, in_channels=3, out_ch=3, ch=128,
                        ch_mult=[1, 2, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
# A tiny UNet with the same topology (odd-factor channel counts 96/192/384 -> head dims 12/24/48 that need
# padding, GroupNorm groups of 3/6/12 channels, 3 attention levels) for fast tests.
TINY_UNET = dict(image_size=32, in_channels=5, out_channels=4, model_channels=96, attention_resolutions=[4, 2, 1],
                 num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                 transformer_depth=1, context_dim=768, use_checkpoint=True, legacy=False)
TINY_DDCONFIG = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=32,
                     ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)

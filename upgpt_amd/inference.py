"""Caller glue of the demo / notebooks around the hot path: the reference's
`ldm/data/generate_utils.py` — `InferenceModel` (:137-189), the person-mask helpers `get_coord` / `get_mask` /
`interp_mask` (:110-134), `get_empty_style` (:103-104), `convert_fname` (:75-96), `load_model_from_config` (:34-49) —
so `app.py` and the inference notebooks drive THIS build through the calls they already make
(`from ldm.data.generate_utils import InferenceModel, convert_fname, interp_mask`).

What is different from the reference, by necessity of the environment rather than by design:

* no torchvision / omegaconf / skimage / pandas imports at module scope (none is needed for these functions; the CLIP
  normalisation constants are applied with torch);
* the CLIP weights: the reference's `clip.load("ViT-L/14")` downloads them; here `InferenceModel(..., clip_weights=path)`
  loads the `clip` package's state dict (one file, both towers) into the two HIP-kernel encoders, and `ckpt=None` /
  `clip_weights=None` leave the corresponding weights at their initial values (tests load recipe weights afterwards);
* the tokenizers (`clip.tokenize`, the hub tokenizer of FrozenCLIPEmbedder) need vocabulary files that are not on disk:
  pass `clip_tokenizer=` / `text_tokenizer=`; without them the string entry points raise RuntimeError.

All compute behind these calls runs through libupk.so (the two CLIP towers, the UNet x DDIM loop, the VAE).
"""
import copy
import re

import numpy as np
import torch

from .config import instantiate_from_config, to_plain

style_names = ['face', 'hair', 'headwear', 'background', 'top', 'outer', 'bottom', 'shoes', 'accesories']  # (sic)

# CLIP pre-processing constants (clip package `_transform`; generate_utils.py:98-101)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
MASK_BG, MASK_FG = -1.0, -0.99215686  # person_mask values: 0/255 and 1/255 mapped to [-1, 1] (deepfashion_inshop.py:235-239)


def load_model_from_config(config, ckpt, verbose=False):
    """generate_utils.py:34-49.  `config` holds a `model` node; `ckpt` is a Lightning checkpoint ({'state_dict': ...}),
    or None for an un-initialised model."""
    cfg = to_plain(config)
    model = instantiate_from_config(cfg["model"] if "model" in cfg else cfg)
    if ckpt is not None:
        print(f"Loading model from {ckpt}")
        pl_sd = torch.load(ckpt, map_location="cpu", weights_only=False)
        if "global_step" in pl_sd:
            print(f"Global Step: {pl_sd['global_step']}")
        missing, unexpected = model.load_state_dict(pl_sd["state_dict"], strict=False)
        if verbose:
            for title, keys in (("missing keys:", missing), ("unexpected keys:", unexpected)):
                if len(keys) > 0:
                    print(title)
                    print(keys)
    model.eval()
    return model


def clip_normalize(img01):
    """[3, H, W] image in [0, 1] -> CLIP-normalised (the Normalize half of generate_utils.clip_transform)."""
    mean = torch.tensor(CLIP_MEAN, dtype=img01.dtype).view(3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=img01.dtype).view(3, 1, 1)
    return (img01 - mean) / std


def get_empty_style():
    """The 'no style' crop: a black 224x224 image through the CLIP transform (generate_utils.py:103-104; float64 like
    the reference's ToTensor of a float64 array)."""
    return clip_normalize(torch.zeros(3, 224, 224, dtype=torch.float64))


def denormalize_style(style):
    """Inverse of the CLIP normalisation for display ([3, H, W] -> [H, W, 3] in [0, 1]); generate_utils.py:52-56."""
    mean = torch.tensor(CLIP_MEAN, dtype=style.dtype).view(3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=style.dtype).view(3, 1, 1)
    return (style * std + mean).clamp(0, 1).permute(1, 2, 0)


def draw_styles(style_batch):
    """Grid of the first eight style crops (generate_utils.py:52-72); needs matplotlib."""
    import matplotlib.pyplot as plt
    fig, axs = plt.subplots(2, 4)
    fig.set_figheight(8)
    fig.set_figwidth(16)
    for i, (name, style) in enumerate(zip(style_names[:-1], style_batch[:-1])):
        ax = axs[i // 4, i % 4]
        ax.imshow(denormalize_style(style.detach().cpu().float()).numpy())
        ax.set_title(name)
        ax.axis('off')
    plt.show()


_FNAME = {"MEN": re.compile(r'MEN(\w+)id(\d+)_(\d)(\w+)'), "WOMEN": re.compile(r'WOMEN(\w+)id(\d+)_(\d)(\w+)')}


def convert_fname(long_name):
    """'fashionWOMENBlouses_Shirtsid0000311501_7additional___fashionWOMEN...01_2side' ->
    ['WOMEN/Blouses_Shirts/id_00003115/01_7_additional', 'WOMEN/Blouses_Shirts/id_00003115/01_2_side']
    (generate_utils.py:75-96: the gender is read from characters 7..9 of the first name and applied to all)."""
    gender = 'MEN' if long_name[7:10] == 'MEN' else 'WOMEN'
    joined = ' '.join(long_name.replace('fashion', '').split('___'))
    return ['%s/%s/id_%s/%s_%s_%s' % (gender, cat, num[:8], num[8:], view, desc)
            for cat, num, view, desc in _FNAME[gender].findall(joined)]


def get_coord(batch_mask):
    """Bounding box [xmin, xmax, ymin, ymax] (rows, then columns) of the non-background part of person_mask
    [1, h, w] (generate_utils.py:110-118).  Like the reference, background (-1) is zeroed IN PLACE when the mask lives
    on the CPU (`.cpu().numpy()` shares memory there)."""
    mask = batch_mask[0].cpu().numpy()
    mask[mask == MASK_BG] = 0
    rows = np.nonzero(np.mean(mask, 1))[0]
    cols = np.nonzero(np.mean(mask, 0))[0]
    return np.array([rows[0], rows[-1], cols[0], cols[-1]])


def get_mask(mask, coord):
    """person_mask of `mask`'s shape/device: background everywhere, foreground inside the inclusive box `coord`
    (generate_utils.py:120-126)."""
    xmin, xmax, ymin, ymax = coord
    new_mask = np.full(tuple(mask.shape), MASK_BG, dtype=mask.cpu().numpy().dtype)
    new_mask[0, xmin:xmax + 1, ymin:ymax + 1] = MASK_FG
    return torch.tensor(new_mask).to(mask.device)


def interp_mask(src_mask, dst_mask, alpha):
    """Box of src and dst blended alpha : (1 - alpha), truncated to int (generate_utils.py:128-134)."""
    coord = (alpha * get_coord(src_mask) + (1 - alpha) * get_coord(dst_mask)).astype(np.int32)
    return get_mask(src_mask, coord)


def load_clip_weights(path, text_encoder=None, image_encoder=None):
    """Loads a state dict of the `clip` package's CLIP model (as `clip.load(name, jit=False)[0].state_dict()` saves it:
    `visual.*`, `transformer.*`, `token_embedding.weight`, `positional_embedding`, `ln_final.*`, `text_projection`,
    `logit_scale`) into the text tower and/or the image tower."""
    sd = torch.load(path, map_location="cpu", weights_only=False)
    sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd.state_dict()
    if text_encoder is not None:
        want = text_encoder.model.state_dict()
        text_encoder.model.load_state_dict({k: sd[k].float() for k in want})
    if image_encoder is not None:
        want = image_encoder.model.visual.state_dict()
        image_encoder.model.visual.load_state_dict({k: sd["visual." + k].float() for k in want})


class InferenceModel:
    """generate_utils.py:137-189.  Holds the two CLIP encoders of the style mixer and the diffusion model whose style
    stage is a pass-through (`mix_style` hands over [9, 768] embeddings)."""

    def __init__(self, config, ckpt, device, clip_weights=None, clip_tokenizer=None, text_tokenizer=None):
        self.device = device
        config = copy.deepcopy(to_plain(config))
        params = config['model']['params']
        text_cfg = {'target': 'ldm.modules.encoders.modules.FrozenCLIPTextEmbedder',
                    'params': {'normalize': False, 'device': device}}
        self.clip_text_encoder = instantiate_from_config(text_cfg)
        self.clip_text_encoder.tokenizer = clip_tokenizer
        style_cfg = dict(params['extra_cond_stages']['style_cond'], params={'device': device})
        self.clip_image_encoder = instantiate_from_config(style_cfg)
        if clip_weights is not None:
            load_clip_weights(clip_weights, self.clip_text_encoder, self.clip_image_encoder)
        self.clip_text_encoder.to(device)
        self.clip_image_encoder.to(device)
        # the diffusion model receives style EMBEDDINGS; its first stage is filled from the checkpoint, not from a file
        params['extra_cond_stages']['style_cond']['target'] = 'ldm.modules.poses.poses.DummyModel'
        params['first_stage_config']['params']['ckpt_path'] = None
        params['cond_stage_config']['params'] = {'device': device}
        self.model = load_model_from_config(config, ckpt).to(device)
        if text_tokenizer is not None and hasattr(self.model.cond_stage_model, "tokenizer"):
            self.model.cond_stage_model.tokenizer = text_tokenizer

    def create_batch(self, batch, repeat=1):
        """One sample -> a batch of `repeat` copies on the device (tensors gain a leading axis, other entries become
        lists); modifies and returns `batch` like the reference (:150-158)."""
        for k, v in batch.items():
            if type(v) == torch.Tensor:
                batch[k] = v.unsqueeze(0).repeat([repeat] + [1] * v.dim()).to(self.device)
            else:
                batch[k] = [v] * repeat
        return batch

    def generate(self, batch, steps=200, repeat=1, use_ema=True):
        """log_images with the demo's settings -> {'reconstruction'?, 'samples'} as float numpy [b, h, w, 3] in [0, 1]
        (:160-170; `repeat` is unused there as well)."""
        with torch.no_grad():
            images = self.model.log_images(batch, ddim_steps=steps, use_ema=use_ema, unconditional_guidance_scale=3.,
                                           unconditional_guidance_label=[""])
        out = {}
        for k, v in images.items():
            v = torch.clamp(v.detach().cpu(), -1., 1.)
            out[k] = v.permute(0, 2, 3, 1).numpy() * 0.5 + 0.5
        return out

    def mix_style(self, s, w, mask=[]):
        """Style embeddings [9, 768] of the crops `s` [9, 3, 224, 224]; slots named in `mask` are blanked (in `s`
        itself, as the reference does) and slots with a text in `w` take the CLIP TEXT embedding instead (:173-189)."""
        slot = {name: i for i, name in enumerate(style_names)}
        for m in mask:
            s[slot[m]] = get_empty_style()
        for k in w:
            if k not in slot:
                raise KeyError("unknown style slot %r (slots: %s)" % (k, ", ".join(style_names)))
        texts = [w.get(name, '') for name in style_names]
        with torch.no_grad():
            image_emb = self.clip_image_encoder(s.unsqueeze(0).to(self.device))
            if any(t != '' for t in texts):
                text_emb = self.clip_text_encoder([texts])
                for i, text in enumerate(texts):
                    if text != '':
                        image_emb[0, i] = text_emb[0, i]
        return image_emb.squeeze(0)

"""Caller glue (upgpt_amd/inference.py == the reference's ldm/data/generate_utils.py surface) against
outputs of the reference's own functions (tests/golden/glue.json, made by tests/golden/make_glue_golden.py)."""
import json
import os
import types

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")
BG, FG = -1.0, -0.99215686


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(G, "glue.json")))


def box_mask(h, w, r0, r1, c0, c1):
    m = torch.full((1, h, w), BG)
    m[0, r0:r1 + 1, c0:c1 + 1] = FG
    return m


def test_import_path_and_names():
    import ldm.data.generate_utils as gu
    from upgpt_amd import inference
    for n in ("InferenceModel", "draw_styles", "convert_fname", "interp_mask", "get_coord", "get_mask", "get_empty_style",
              "load_model_from_config", "style_names"):
        assert getattr(gu, n) is getattr(inference, n)
    assert gu.style_names == ['face', 'hair', 'headwear', 'background', 'top', 'outer', 'bottom', 'shoes', 'accesories']


def test_convert_fname(golden):
    from ldm.data.generate_utils import convert_fname
    assert len(golden["fnames"]) >= 6
    for s, want in golden["fnames"].items():
        assert convert_fname(s) == want, s


def test_mask_helpers(golden):
    from ldm.data.generate_utils import get_coord, get_mask, interp_mask
    for c in golden["coords"]:
        assert [int(v) for v in get_coord(box_mask(32, 24, *c["box"]))] == c["coord"] == list(c["box"])
    for c in golden["interp"]:
        m = interp_mask(box_mask(32, 24, *c["src"]), box_mask(32, 24, *c["dst"]), c["alpha"])
        assert list(m.shape) == c["shape"] and str(m.dtype) == c["dtype"]
        assert sorted(set(round(float(v), 8) for v in m.unique())) == c["values"]
        fg = (m[0] != BG).nonzero()
        box = [int(fg[:, 0].min()), int(fg[:, 0].max()), int(fg[:, 1].min()), int(fg[:, 1].max())]
        assert box == c["box"] and len(fg) == c["n_fg"]
        assert torch.equal(m, get_mask(torch.zeros(1, 32, 24), box))
    # the synthetic bench mask (upgpt_amd/synth.py) is such a mask: its box survives a round trip
    from upgpt_amd import synth
    pm = synth.person_mask(1, 32, 24)[0]
    assert torch.equal(get_mask(pm, get_coord(pm.clone())), pm)


def test_empty_style_is_the_normalised_black_image():
    from ldm.data.generate_utils import get_empty_style
    e = get_empty_style()
    assert e.shape == (3, 224, 224) and e.dtype == torch.float64
    want = -np.array([0.48145466, 0.4578275, 0.40821073]) / np.array([0.26862954, 0.26130258, 0.27577711])
    assert np.allclose(e[:, 0, 0].numpy(), want, atol=1e-12) and float((e - e[:, :1, :1]).abs().max()) == 0.0
    s = torch.zeros(9, 3, 224, 224)
    s[2] = e  # what mix_style does with a masked slot: cast to the crops' dtype
    assert s.dtype == torch.float32 and abs(float(s[2, 0, 0, 0]) - want[0]) < 1e-6


def test_create_batch_and_generate_postprocessing(golden):
    from ldm.data.generate_utils import InferenceModel
    holder = types.SimpleNamespace(device="cpu")
    batch = {"image": torch.arange(24.).view(2, 4, 3), "txt": "a person", "smpl": torch.ones(1, 85), "fname": "x.jpg"}
    got = InferenceModel.create_batch(holder, batch, repeat=3)
    assert got is batch
    g = golden["create_batch"]
    for k, v in got.items():
        assert (list(v.shape) if torch.is_tensor(v) else v) == g[k], k
    assert float(got["image"].sum()) == g["image_sum"]

    class M:
        def log_images(self, batch, **kw):
            self.kw = kw
            gen = torch.Generator().manual_seed(1)
            return {"samples": torch.randn(2, 3, 4, 5, generator=gen) * 1.5}

    holder = types.SimpleNamespace(device="cpu", model=M())
    img = InferenceModel.generate(holder, {}, steps=7, use_ema=False)
    p = golden["generate_post"]
    assert list(img["samples"].shape) == p["shape"] and isinstance(img["samples"], np.ndarray)
    assert abs(float(img["samples"].sum()) - p["sum"]) < 1e-4
    assert float(img["samples"].min()) == p["min"] == 0.0 and float(img["samples"].max()) == p["max"] == 1.0
    assert holder.model.kw == p["kwargs"]


def test_mix_style_rejects_unknown_slot_and_needs_gpu_for_compute():
    from ldm.data.generate_utils import InferenceModel
    holder = types.SimpleNamespace(device="cpu")
    with pytest.raises(KeyError):
        InferenceModel.mix_style(holder, torch.zeros(9, 3, 224, 224), {"hat": "a red hat"})


# ---------------------------------------------------------------------------------------------------------------- GPU
def _fake_clip_tokenize(texts):
    """Stand-in for clip.tokenize (its BPE vocabulary is not available offline): [n, 77] ids, <start> ... <end> 0 0 ..."""
    if isinstance(texts, str):
        texts = [texts]
    ids = torch.zeros(len(texts), 77, dtype=torch.long)
    for r, t in enumerate(texts):
        body = [1000 + (ord(c) * 37) % 40000 for c in t][:75]
        ids[r, 0] = 49406
        ids[r, 1:1 + len(body)] = torch.tensor(body, dtype=torch.long) if body else ids[r, 1:1]
        ids[r, 1 + len(body)] = 49407
    return ids


class _FakeHFTokenizer:
    """Stand-in for transformers' CLIPTokenizer call as FrozenCLIPEmbedder makes it (modules.py:152-155)."""

    def __call__(self, text, truncation=True, max_length=77, padding="max_length", return_tensors="pt", **kw):
        ids = _fake_clip_tokenize(text)
        ids[ids == 0] = 49407  # the hub tokenizer pads with <end>
        return {"input_ids": ids}


@pytest.mark.gpu
def test_inference_model_end_to_end():
    """app.py's sequence (app.py:267-272): mix_style -> create_batch -> generate, through InferenceModel with both CLIP
    encoders and the text cond stage on the HIP kernels (tiny UNet/VAE, full-size ViT-L/14 towers, recipe weights)."""
    import copy
    from ldm.data.generate_utils import InferenceModel, get_empty_style
    from upgpt_amd import synth
    from upgpt_amd.config import load_config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = copy.deepcopy(load_config(os.path.join(root, "configs", "upgpt_bbox_model.yaml")))
    p = cfg["model"]["params"]
    p["unet_config"]["params"].update(synth.TINY_UNET)
    p["first_stage_config"]["params"]["ddconfig"].update(synth.TINY_DDCONFIG)
    # the reference's bbox.yaml stages (bbox.yaml:81-93)
    p["cond_stage_config"] = {"target": "ldm.modules.encoders.modules.FrozenCLIPEmbedder"}
    p["extra_cond_stages"]["style_cond"]["target"] = "ldm.modules.encoders.modules.FrozenClipImageEmbedder2"
    keep = copy.deepcopy(cfg)
    im = InferenceModel(cfg, None, "cuda", clip_tokenizer=_fake_clip_tokenize, text_tokenizer=_FakeHFTokenizer())
    assert cfg == keep, "the caller's config must not be modified"
    assert type(im.model.extra_cond_models[0]).__name__ == "DummyModel"
    assert type(im.model.cond_stage_model).__name__ == "FrozenCLIPEmbedder"
    synth.fill_module_(im.model, prefixes=("model.diffusion_model.", "first_stage_model.", "extra_cond_models.",
                                           "cond_stage_model."))
    for enc, pre in ((im.clip_text_encoder, "clip_text_encoder."), (im.clip_image_encoder, "extra_cond_models.0.")):
        enc.load_state_dict({k: synth.synth_tensor(pre + k, tuple(v.shape)) for k, v in enc.state_dict().items()})
    g0 = torch.Generator().manual_seed(11)
    styles = torch.randn(9, 3, 224, 224, generator=g0)
    plain = im.clip_image_encoder(styles.unsqueeze(0).cuda())[0]
    emb = im.mix_style(styles, {"top": "a red shirt"}, mask=["hair"])
    assert emb.shape == (9, 768) and torch.isfinite(emb).all()
    assert torch.equal(styles[1], get_empty_style().float())  # the masked slot was blanked in the caller's tensor
    text = im.clip_text_encoder([["", "", "", "", "a red shirt", "", "", "", ""]])[0]
    blank = im.clip_image_encoder(get_empty_style().float()[None, None].cuda())[0, 0]
    for i in range(9):
        want = text[4] if i == 4 else (blank if i == 1 else plain[i])
        assert torch.allclose(emb[i], want, atol=2e-2, rtol=1e-2), i  # (N=1 and N=9 programs use different GEMM tilings)
    assert not torch.allclose(emb[4], plain[4], atol=1e-2)
    # no text: the text tower (and its tokenizer) is not needed
    im.clip_text_encoder.tokenizer = None
    assert torch.allclose(im.mix_style(styles, {}), im.clip_image_encoder(styles.unsqueeze(0).cuda())[0])
    batch = {"image": torch.rand(256, 192, 3, generator=g0) * 2 - 1, "txt": "a woman in a red shirt", "styles": emb,
             "smpl": 0.5 * torch.randn(1, 85, generator=g0), "person_mask": synth.person_mask(1, 32, 24)[0]}
    batch = im.create_batch(batch, repeat=2)
    assert batch["styles"].shape == (2, 9, 768) and batch["txt"] == ["a woman in a red shirt"] * 2
    assert batch["person_mask"].shape == (2, 1, 32, 24) and batch["image"].device.type == "cuda"
    out = im.generate(batch, steps=4)
    assert set(out) == {"reconstruction", "samples"}
    for k in out:
        assert out[k].shape == (2, 256, 192, 3) and np.isfinite(out[k]).all()
        assert out[k].min() >= 0.0 and out[k].max() <= 1.0
    assert out["samples"].std() > 0.01
    # the text reaches the UNet: another prompt, same seed -> another image
    torch.manual_seed(3)
    a = im.generate(dict(batch), steps=2)["samples"]
    torch.manual_seed(3)
    b = im.generate(dict(batch, txt=["a man in a blue coat"] * 2), steps=2)["samples"]
    torch.manual_seed(3)
    a2 = im.generate(dict(batch), steps=2)["samples"]
    assert np.array_equal(a, a2) and not np.allclose(a, b, atol=1e-3)

"""Golden vectors for the CLIP image tower.  The reference's FrozenClipImageEmbedder2
(ldm/modules/encoders/modules.py:234-256) calls `encode_image` of OpenAI's `clip` package model ViT-L/14 (third-party,
git main, not installed here).  The same VisionTransformer is implemented by transformers' CLIPVisionModelWithProjection
(available offline, random init): this script maps recipe weights (upgpt_amd/synth.py, keyed by the clip package's
parameter names under the checkpoint prefix) onto it, runs it on CPU fp32 on seeded inputs and stores the image
embeddings.  Only data is committed.

    python tests/golden/make_clip_image_golden.py       ->  tests/golden/clip_image.npz
"""
import importlib.util
import os
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, "..", "..", "upgpt_amd", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


synth = load("synth")
from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection  # noqa: E402
import transformers  # noqa: E402

PREFIX = "extra_cond_models.0.model.visual."  # style_cond is the first extra conditioning stage (bbox.yaml:84-93)
W, LAYERS, HEADS, PATCH, IMG, OUT = 1024, 24, 16, 14, 224, 768
cfg = CLIPVisionConfig(hidden_size=W, intermediate_size=4 * W, num_hidden_layers=LAYERS, num_attention_heads=HEADS,
                       image_size=IMG, patch_size=PATCH, projection_dim=OUT, hidden_act="quick_gelu", layer_norm_eps=1e-5)
model = CLIPVisionModelWithProjection(cfg).eval()


def rec(key, shape):
    return synth.synth_tensor(PREFIX + key, shape)


n_tok = (IMG // PATCH) ** 2 + 1
new = {
    "vision_model.embeddings.class_embedding": rec("class_embedding", (W,)),
    "vision_model.embeddings.patch_embedding.weight": rec("conv1.weight", (W, 3, PATCH, PATCH)),
    "vision_model.embeddings.position_embedding.weight": rec("positional_embedding", (n_tok, W)),
    "vision_model.pre_layrnorm.weight": rec("ln_pre.weight", (W,)),
    "vision_model.pre_layrnorm.bias": rec("ln_pre.bias", (W,)),
    "vision_model.post_layernorm.weight": rec("ln_post.weight", (W,)),
    "vision_model.post_layernorm.bias": rec("ln_post.bias", (W,)),
    "visual_projection.weight": rec("proj", (W, OUT)).t().contiguous(),
}
n_keys = 8
for i in range(LAYERS):
    b = "transformer.resblocks.%d." % i
    h = "vision_model.encoder.layers.%d." % i
    wq, bq = rec(b + "attn.in_proj_weight", (3 * W, W)), rec(b + "attn.in_proj_bias", (3 * W,))
    for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
        new[h + "self_attn.%s.weight" % nm] = wq[j * W:(j + 1) * W].clone()
        new[h + "self_attn.%s.bias" % nm] = bq[j * W:(j + 1) * W].clone()
    for src, dst in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"),
                     ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
        for wb in ("weight", "bias"):
            new[h + dst + "." + wb] = rec(b + src + "." + wb, tuple(model.state_dict()[h + dst + "." + wb].shape))
    n_keys += 12
missing, unexpected = model.load_state_dict(new, strict=False)
assert not unexpected and not [m for m in missing if "position_ids" not in m], (missing, unexpected)
g = torch.Generator(device="cpu").manual_seed(777)
x = torch.randn(1, 3, 3, IMG, IMG, generator=g)  # [b, n crops, 3, 224, 224], already "pre-processed"
with torch.no_grad():
    out = model(pixel_values=x.reshape(3, 3, IMG, IMG)).image_embeds.reshape(1, 3, OUT)
np.savez_compressed(os.path.join(HERE, "clip_image.npz"), image_embeds=out.numpy().astype(np.float32),
                    x_crc=np.uint32(zlib.crc32(x.numpy().tobytes()) & 0xFFFFFFFF), abs_mean=np.float32(out.abs().mean()),
                    transformers_version=np.bytes_(transformers.__version__), n_keys=np.int32(n_keys))
print("wrote clip_image.npz: out", tuple(out.shape), "abs mean %.4f" % out.abs().mean(), "keys", n_keys)

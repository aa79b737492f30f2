"""Golden vectors for FrozenCLIPTextEmbedder (the text encoder InferenceModel.mix_style uses).  The
reference's class (ldm/modules/encoders/modules.py:164-198) calls `encode_text` of OpenAI's `clip` package model
ViT-L/14 (third-party, git main, not installed here): token + positional embedding -> 12 causal pre-LN blocks ->
ln_final -> the end-of-text row (argmax of the token ids) @ text_projection.  The same network is implemented by
transformers' CLIPTextModelWithProjection (available offline, random init): this script maps recipe weights
(upgpt_amd/synth.py, keyed by the clip package's parameter names under `clip_text_encoder.model.`) onto it, runs it on
CPU fp32 on seeded token ids and stores the text embeddings.  Only data is committed.

    python tests/golden/make_clip_textproj_golden.py    ->  tests/golden/clip_textproj.npz
"""
import importlib.util
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("synth", os.path.join(HERE, "..", "..", "upgpt_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synth)

from transformers import CLIPTextConfig, CLIPTextModelWithProjection  # noqa: E402
import transformers  # noqa: E402

PREFIX = "clip_text_encoder.model."
W, F, LAYERS, HEADS, S, V, OUT = 768, 3072, 12, 12, 77, 49408, 768
cfg = CLIPTextConfig(vocab_size=V, hidden_size=W, intermediate_size=F, num_hidden_layers=LAYERS, num_attention_heads=HEADS,
                     max_position_embeddings=S, hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=OUT,
                     eos_token_id=2)  # eos_token_id 2 = the argmax pooling of the original CLIP (encode_text)
model = CLIPTextModelWithProjection(cfg).eval()


def rec(key, shape):
    return synth.synth_tensor(PREFIX + key, shape)


msd = model.state_dict()
new = {
    "text_model.embeddings.token_embedding.weight": rec("token_embedding.weight", (V, W)),
    "text_model.embeddings.position_embedding.weight": rec("positional_embedding", (S, W)),
    "text_model.final_layer_norm.weight": rec("ln_final.weight", (W,)),
    "text_model.final_layer_norm.bias": rec("ln_final.bias", (W,)),
    "text_projection.weight": rec("text_projection", (W, OUT)).t().contiguous(),
}
n_keys = 5
for i in range(LAYERS):
    b = "transformer.resblocks.%d." % i
    h = "text_model.encoder.layers.%d." % i
    wq, bq = rec(b + "attn.in_proj_weight", (3 * W, W)), rec(b + "attn.in_proj_bias", (3 * W,))
    for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
        new[h + "self_attn.%s.weight" % nm] = wq[j * W:(j + 1) * W].clone()
        new[h + "self_attn.%s.bias" % nm] = bq[j * W:(j + 1) * W].clone()
    for src, dst in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"),
                     ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
        for wb in ("weight", "bias"):
            new[h + dst + "." + wb] = rec(b + src + "." + wb, tuple(msd[h + dst + "." + wb].shape))
    n_keys += 12
# some transformers versions keep the tower without the text_model. prefix
if not any(k.startswith("text_model.") for k in msd):
    new = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in new.items()}
missing, unexpected = model.load_state_dict(new, strict=False)
assert not unexpected and not [m for m in missing if "position_ids" not in m], (missing, unexpected)
g = torch.Generator(device="cpu").manual_seed(2024)
ids = torch.randint(0, 49000, (9, S), generator=g)
ids[:, 0] = 49406                                  # <|startoftext|>
for r, n in enumerate((2, 5, 9, 14, 20, 33, 50, 76, 3)):
    ids[r, n] = 49407                              # <|endoftext|> (the highest id: argmax finds it)
    ids[r, n + 1:] = 0                             # clip.tokenize pads with zeros
with torch.no_grad():
    o = model(input_ids=ids)
    out = o.text_embeds
    # cross-check with the formula of clip/model.py encode_text on the same hidden states
    alt = o.last_hidden_state[torch.arange(9), ids.argmax(-1)] @ model.text_projection.weight.t()
assert torch.allclose(out, alt, atol=1e-5), float((out - alt).abs().max())
np.savez_compressed(os.path.join(HERE, "clip_textproj.npz"), ids=ids.numpy().astype(np.int32),
                    text_embeds=out.numpy().astype(np.float32), abs_mean=np.float32(out.abs().mean()),
                    transformers_version=np.bytes_(transformers.__version__), n_keys=np.int32(n_keys))
print("wrote clip_textproj.npz: out", tuple(out.shape), "abs mean %.4f" % out.abs().mean(), "keys", n_keys)

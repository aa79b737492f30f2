"""Golden vectors for the CLIP text tower.  The reference's FrozenCLIPEmbedder is Hugging Face's
CLIPTextModel (ldm/modules/encoders/modules.py:137-162; transformers 4.19.2 pinned in its environment.yaml — a
third-party dependency, not vendored); this script runs THAT implementation (transformers as installed in the build
container) on CPU fp32 with recipe weights (upgpt_amd/synth.py: a pure function of key name and shape) and seeded
token ids, and stores inputs + last_hidden_state.  Only data is committed.

    python tests/golden/make_clip_text_golden.py        ->  tests/golden/clip_text.npz
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("synth", os.path.join(HERE, "..", "..", "upgpt_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synth)

from transformers import CLIPTextConfig, CLIPTextModel  # noqa: E402
import transformers  # noqa: E402

PREFIX = "cond_stage_model.transformer."  # where the reference's checkpoints keep the tower
cfg = CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                     num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                     projection_dim=768)
model = CLIPTextModel(cfg).eval()
sd = model.state_dict()
new = {}
for k, v in sd.items():
    ck = k if k.startswith("text_model.") else "text_model." + k  # 4.x checkpoints carry the text_model. prefix
    if v.dtype.is_floating_point:
        new[k] = synth.synth_tensor(PREFIX + ck, tuple(v.shape))
missing, unexpected = model.load_state_dict(new, strict=False)
assert not unexpected
g = torch.Generator(device="cpu").manual_seed(4242)
ids = torch.randint(0, 49408, (2, 77), generator=g)
ids[0, 0], ids[1, 0] = 49406, 49406       # <|startoftext|>
ids[0, 20:] = 49407                        # <|endoftext|> padding, as the tokenizer produces
with torch.no_grad():
    out = model(input_ids=ids).last_hidden_state
np.savez_compressed(os.path.join(HERE, "clip_text.npz"), ids=ids.numpy().astype(np.int32),
                    last_hidden_state=out.numpy().astype(np.float16),
                    abs_mean=np.float32(out.abs().mean()), transformers_version=np.bytes_(transformers.__version__),
                    n_keys=np.int32(len(new)))
print("wrote clip_text.npz: out", tuple(out.shape), "abs mean %.4f" % out.abs().mean(), "keys", len(new),
      "transformers", transformers.__version__)

"""ldm.modules.encoders.modules — conditioning encoders.

* FrozenCLIPEmbedder (modules.py:137-162): the CLIP text tower runs on the HIP kernels
  (upgpt_amd/clip_text.py); its tokenizer files are not available offline, so `encode(text)` needs them on disk
  and `encode_tokens(ids)` takes token ids.
* The CLIP image embedder (modules.py:234-256) is not built yet: the name resolves so that bbox.yaml instantiates;
  calling it explains what to do instead — feed precomputed embeddings through DummyModel, exactly as the
  reference's own InferenceModel does (ldm/data/generate_utils.py:142-144)."""
from torch import nn

from upgpt_amd.clip_text import CLIPTextTransformer, FrozenCLIPEmbedder  # noqa: F401


class _ExternalEncoder(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, *a, **k):
        raise NotImplementedError(
            "%s (CLIP image tower) is not part of upgpt_amd yet: pass precomputed [B, 9, 768] embeddings and set the "
            "stage's target to ldm.modules.poses.poses.DummyModel" % type(self).__name__)

    encode = forward


class FrozenClipImageEmbedder2(_ExternalEncoder):
    pass


class FrozenCLIPTextEmbedder(_ExternalEncoder):
    pass

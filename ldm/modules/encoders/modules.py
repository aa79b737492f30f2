"""ldm.modules.encoders.modules — the CLIP conditioning encoders on the HIP kernels.

* FrozenCLIPEmbedder (modules.py:137-162): CLIP text tower (upgpt_amd/clip_text.py); its tokenizer files are not
  available offline, so `encode(text)` needs them on disk and `encode_tokens(ids)` takes token ids.
* FrozenClipImageEmbedder2 (modules.py:234-256): CLIP ViT-L/14 image tower (upgpt_amd/clip_image.py),
  [b, n, 3, 224, 224] pre-processed crops -> [b, n, 768].
* FrozenCLIPTextEmbedder (modules.py:164-198): the `clip`-package text encoder InferenceModel.mix_style uses
  (upgpt_amd/clip_text.py, OpenAI key layout, end-of-text pooling + text_projection -> [n, 768])."""
from upgpt_amd.clip_image import CLIPVisual, FrozenClipImageEmbedder2  # noqa: F401
from upgpt_amd.clip_text import (CLIPTextTower, CLIPTextTransformer, FrozenCLIPEmbedder,  # noqa: F401
                                 FrozenCLIPTextEmbedder)

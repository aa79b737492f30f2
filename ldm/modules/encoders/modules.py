"""ldm.modules.encoders.modules — the CLIP conditioning encoders are OUTSIDE the hot path
(next row; their weights are downloaded at run time by the reference and
are not available offline).  The names resolve so that bbox.yaml instantiates; calling one
explains what to do instead: feed precomputed embeddings through DummyModel, exactly as
the reference's own InferenceModel does (ldm/data/generate_utils.py:142-144)."""
from torch import nn


class _ExternalEncoder(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, *a, **k):
        raise NotImplementedError(
            "%s (CLIP) is not part of upgpt_amd yet: pass precomputed [B, 77|9, 768] embeddings and set the "
            "stage's target to ldm.modules.poses.poses.DummyModel" % type(self).__name__)

    encode = forward


class FrozenCLIPEmbedder(_ExternalEncoder):
    pass


class FrozenClipImageEmbedder2(_ExternalEncoder):
    pass


class FrozenCLIPTextEmbedder(_ExternalEncoder):
    pass

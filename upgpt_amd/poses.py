"""Conditioning stages that sit next to (not on) the hot path.

* LinearProject: the SMPL pose vector (85 floats: 72 body pose + 10 betas + 3 camera) becomes one
  768-wide context token.  The sub-module MUST be called `model`: reference checkpoints store it as
  `extra_cond_models.1.model.{weight,bias}` (ldm/modules/poses/poses.py:3-9, bbox.yaml:88-93).
  It runs once per batch, outside the sampling loop, so it stays a plain torch Linear.
* DummyModel: pass-through used when embeddings are computed outside the model — the reference's
  own InferenceModel swaps it in for both CLIP stages (ldm/data/generate_utils.py:142-144).
"""
import torch
from torch import nn


class LinearProject(nn.Module):
    def __init__(self, input_dim, output_dim):
        nn.Module.__init__(self)
        self.input_dim, self.output_dim = int(input_dim), int(output_dim)
        self.model = nn.Linear(self.input_dim, self.output_dim, bias=True)

    def forward(self, smpl):
        if smpl.shape[-1] != self.input_dim:
            raise ValueError("LinearProject expects [..., %d] SMPL vectors, got %s" % (self.input_dim,
                                                                                   tuple(smpl.shape)))
        return torch.nn.functional.linear(smpl, self.model.weight, self.model.bias)


class DummyModel(nn.Identity):
    """Accepts and ignores any constructor arguments (configs pass CLIP kwargs)."""

    def __init__(self, *unused_args, **unused_kwargs):
        nn.Identity.__init__(self)

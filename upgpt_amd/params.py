"""Parameter containers whose state_dict() keys equal the reference's.

The reference's checkpoints are Lightning state-dicts keyed by the nn.Module attribute
paths of its constructors (SURVEY.md §5 'Checkpoint / resume', §8b-3), e.g.
`model.diffusion_model.input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight`.
ParamTree materialises exactly those paths from an arch.* shape table; it holds fp32
master weights only — there is deliberately NO forward(): compute happens in the HIP
engine, and a CPU tensor reaching it is an error, not a fallback.
"""
import torch
from torch import nn


class ParamNode(nn.Module):
    """Anonymous interior node of a ParamTree (digit-named children are fine, as in
    nn.Sequential / nn.ModuleList)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("ParamNode holds weights only; use the owning model's forward")


class ParamTree(ParamNode):
    def __init__(self, shapes=None, dtype=torch.float32):
        super().__init__()
        if shapes:
            self.add_params(shapes, dtype)

    def add_params(self, shapes, dtype=torch.float32):
        for name, shape in shapes.items():
            parts = name.split(".")
            node = self
            for part in parts[:-1]:
                child = node._modules.get(part)
                if child is None:
                    child = ParamNode()
                    node.add_module(part, child)
                node = child
            node.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape, dtype=dtype), requires_grad=True))
        return self


def weights_fingerprint(module):
    """Cheap change detector for packed-weight caches: in-place updates bump
    Tensor._version, replacement / device moves change data_ptr.  Writes through `.data` (p.data.copy_(...)) bump
    neither: after such an edit call the owning model's invalidate() (LitEma.copy_to / restore write through
    torch.no_grad() + copy_ for exactly that reason)."""
    v, ptr = 0, 0
    for p in module.parameters():
        v += p._version
        ptr ^= p.data_ptr()
    for b in module.buffers():
        v += b._version
        ptr ^= b.data_ptr()
    return (v, ptr)

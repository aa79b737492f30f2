"""Golden vectors for the caller glue, produced by the REAL reference functions in
/root/reference/ldm/data/generate_utils.py (imported read-only in the build container): `get_coord`, `get_mask`,
`interp_mask`, `convert_fname`, `InferenceModel.create_batch` and the post-processing half of
`InferenceModel.generate`.  Only inputs/outputs are committed (tests/golden/glue.json).

The module imports torchvision, omegaconf, skimage, pandas, matplotlib and two dataset modules at import time; none of
them is used by the functions above, so inert stand-in modules are registered for the import only.  (`get_empty_style`
depends on torchvision's transforms and is therefore NOT pinned by this script: its value, -mean/std of the CLIP
normalisation, is checked analytically in the test.)

    python tests/golden/make_glue_golden.py     ->  tests/golden/glue.json
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != ROOT]  # the repo's `ldm` alias must not shadow


class _Inert(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = sys.modules.get(self.__name__ + "." + name)
        return sub if sub is not None else (lambda *a, **k: None)


for name in ("torchvision", "torchvision.transforms", "omegaconf", "skimage", "skimage.metrics", "pandas",
             "matplotlib", "matplotlib.pyplot", "ldm.data.deepfashion_inshop", "ldm.data.segm_utils", "tqdm", "PIL"):
    if name not in sys.modules:
        sys.modules[name] = _Inert(name)
import ldm.data.generate_utils as gu  # noqa: E402

assert gu.__file__.startswith(REF)
BG, FG = -1.0, -0.99215686


def box_mask(h, w, r0, r1, c0, c1):
    m = torch.full((1, h, w), BG)
    m[0, r0:r1 + 1, c0:c1 + 1] = FG
    return m


out = {"fnames": {}, "coords": [], "interp": [], "create_batch": {}, "generate_post": {}}
for s in ("fashionWOMENBlouses_Shirtsid0000311501_7additional___fashionWOMENBlouses_Shirtsid0000311501_2side",
          "fashionWOMENShortsid0000478403_4full___fashionWOMENShortsid0000478403_1front",
          "fashionMENTees_Tanksid0000481201_1front", "fashionWOMENDressesid0000676702_4full",
          "fashionMENTees_Tanksid0000260306_1front___fashionWOMENDressesid0000465104_4full", "nothing_matches_here"):
    out["fnames"][s] = gu.convert_fname(s)
boxes = [(4, 27, 6, 17), (0, 31, 0, 23), (10, 10, 3, 3), (2, 20, 12, 23), (16, 30, 0, 5)]
for b in boxes:
    out["coords"].append({"box": b, "coord": [int(v) for v in gu.get_coord(box_mask(32, 24, *b))]})
for a, b, alpha in ((boxes[0], boxes[3], 0.25), (boxes[0], boxes[3], 0.5), (boxes[4], boxes[1], 0.8),
                    (boxes[2], boxes[0], 0.0), (boxes[2], boxes[0], 1.0), (boxes[3], boxes[4], 0.333)):
    m = gu.interp_mask(box_mask(32, 24, *a), box_mask(32, 24, *b), alpha)
    fg = (m[0] != BG).nonzero()
    out["interp"].append({"src": a, "dst": b, "alpha": alpha, "shape": list(m.shape), "dtype": str(m.dtype),
                          "values": sorted(set(round(float(v), 8) for v in m.unique())),
                          "box": [int(fg[:, 0].min()), int(fg[:, 0].max()), int(fg[:, 1].min()), int(fg[:, 1].max())],
                          "n_fg": int(len(fg))})
# create_batch only touches self.device
holder = types.SimpleNamespace(device="cpu")
batch = {"image": torch.arange(24.).view(2, 4, 3), "txt": "a person", "smpl": torch.ones(1, 85), "fname": "x.jpg"}
got = gu.InferenceModel.create_batch(holder, dict(batch), repeat=3)
out["create_batch"] = {k: (list(v.shape) if torch.is_tensor(v) else v) for k, v in got.items()}
out["create_batch"]["image_sum"] = float(got["image"].sum())


# the post-processing of generate() on a fixed tensor (log_images replaced by a constant)
class _M:
    def log_images(self, batch, **kw):
        self.kw = kw
        g = torch.Generator().manual_seed(1)
        return {"samples": torch.randn(2, 3, 4, 5, generator=g) * 1.5}


holder = types.SimpleNamespace(device="cpu", model=_M())
img = gu.InferenceModel.generate(holder, {}, steps=7, use_ema=False)
out["generate_post"] = {"shape": list(img["samples"].shape), "sum": float(img["samples"].sum()),
                        "min": float(img["samples"].min()), "max": float(img["samples"].max()),
                        "kwargs": {k: v for k, v in holder.model.kw.items()}}
json.dump(out, open(os.path.join(HERE, "glue.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out)[:600])

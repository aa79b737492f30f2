"""Times the CLIP conditioning encoders for one bench batch (8 prompts, 8 x 9 style crops). Dev tool."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upgpt_amd import synth
from upgpt_amd.clip_image import FrozenClipImageEmbedder2
from upgpt_amd.clip_text import FrozenCLIPEmbedder
from upgpt_amd.engine import TUNE_CACHE

img = FrozenClipImageEmbedder2()
img.load_state_dict({k: synth.synth_tensor("extra_cond_models.0." + k, tuple(v.shape)) for k, v in img.state_dict().items()})
img = img.cuda()
txt = FrozenCLIPEmbedder()
txt.load_state_dict({k: synth.synth_tensor("cond_stage_model." + k, tuple(v.shape)) for k, v in txt.state_dict().items()})
txt = txt.cuda()
x = torch.randn(8, 9, 3, 224, 224).cuda()
ids = torch.randint(0, 49408, (8, 77))
for name, fn in (("image tower, 72 crops", lambda: img(x)), ("text tower, 8 prompts", lambda: txt.encode_tokens(ids))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    print("%-26s %.2f ms" % (name, (time.perf_counter() - t0) / 3 * 1e3))
if len(sys.argv) > 1:
    TUNE_CACHE.save(sys.argv[1])

"""Re-tunes the GEMM launches of the CLIP image tower (72 crops = one bench batch of 8 x 9 style crops) over every
configuration, incl. the big-tile family, with real activations in the plan's buffers; writes the merged tuning cache.

    python scripts/tune_clip.py [out.json] [n_images ...]
"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upgpt_amd import synth
from upgpt_amd.clip_image import FrozenClipImageEmbedder2
from upgpt_amd.engine import TUNE_CACHE

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tuned_gfx950.json"
counts = [int(v) for v in sys.argv[2:]] or [72]
img = FrozenClipImageEmbedder2()
img.load_state_dict({k: synth.synth_tensor("extra_cond_models.0." + k, tuple(v.shape)) for k, v in img.state_dict().items()})
img = img.cuda()
vis = img.model.visual


def tower_ms(x, n=3):
    img(x); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): img(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for N in counts:
    x = torch.randn(N // 9 if N % 9 == 0 else 1, 9 if N % 9 == 0 else N, 3, 224, 224).cuda()
    before = tower_ms(x)
    plan = vis._plans[N]
    seen = set()
    for d, key in plan.convs:
        if key in seen:
            continue
        seen.add(key)
        old = TUNE_CACHE.get(key)
        cfg, sk, best_us, dflt_us = plan.ctx.conv_autotune(d, 3)
        print("%-52s %s sk %d  %.1f us (was %s)" % (key, plan.ctx.lib.upk_conv_config_name(cfg).decode(), sk, best_us, old), flush=True)
        TUNE_CACHE.put(key, cfg, sk, best_us, dflt_us)
    TUNE_CACHE.save(out)
    vis._plans.clear()
    print("image tower, %d crops: %.2f -> %.2f ms" % (N, before, tower_ms(x)), flush=True)
print("entries:", len(TUNE_CACHE.d), "->", out)

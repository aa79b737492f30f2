"""Autotunes every conv/GEMM launch of the bench workloads on the GPU and writes the tuning
cache (copy gpurun_out/tuned_gfx950.json to upgpt_amd/tuned_gfx950.json and commit).

    UPGPT_AUTOTUNE=1 python scripts/tune.py [out.json]
"""
import contextlib
import io
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["UPGPT_AUTOTUNE"] = "1"
# time every candidate with flushed caches (the in-forward condition), unless UPK_TUNE_WARM=1
if os.environ.get("UPK_TUNE_WARM", "0") != "1":
    os.environ.setdefault("UPK_TUNE_COLD", "1")
os.environ.setdefault("UPGPT_TUNE_REPS", "8")
import upgpt_amd  # noqa: E402
from upgpt_amd import synth  # noqa: E402
from upgpt_amd.engine import TUNE_CACHE  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tuned_gfx950.json"
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["bbox"]
if os.environ.get("UPGPT_TUNE_KEEP", "0") != "1":
    TUNE_CACHE.d = {}  # re-measure everything (kernel set may have changed)


def summarize(name, emitter):
    best = dflt = 0.0
    for d, key in emitter.convs:
        e = TUNE_CACHE.get(key)
        best += e[2]
        dflt += e[3]
    print("%-40s %4d convs: cost-model %.0f us -> tuned %.0f us" % (name, len(emitter.convs), dflt, best), flush=True)


for kind in kinds:
    with contextlib.redirect_stdout(io.StringIO()):
        model = upgpt_amd.build_model("bbox" if kind in ("bbox_cfg", "bbox_small") else ("upscale" if kind == "upscale_true" else kind))
    synth.fill_module_(model)
    model = model.cuda()
    unet = model.model.diffusion_model
    C = model.channels
    ntok = 86 if kind in ("upscale", "upscale_true") else 87
    shapes = [(8, 32, 32, 50), (8, 32, 24, 50)] if kind == "bbox" else [(4, 64, 64, 50)]
    if kind == "bbox_cfg":  # classifier-free guidance runs the UNet on 2*B rows
        shapes, kind = [(16, 32, 32, 50), (16, 32, 24, 50)], "bbox"
    if kind == "upscale_true":  # BASELINE configs[4] at the size the reference's config states: latent 3 x 128 x 96, bs 4
        shapes, kind = [(4, 128, 96, 50)], "upscale"
    if kind == "bbox_small":  # the demo's batch sizes (app.py: 1 sample, interpolation rows of 2-4), 256x192
        shapes, kind = [(1, 32, 24, 50), (2, 32, 24, 50), (4, 32, 24, 50)], "bbox"
    for (B, H, W, S) in shapes:
        # LayerNorm folded into its consumer GEMM / separate, ResBlock skip projection appended to the second conv /
        # separate: every variant gets measured, the emitter then picks per shape
        for fold, skf in (("1", "0"), ("0", "0"), (None, "1")):
            if fold is None:
                os.environ.pop("UPGPT_LN_FOLD", None)
            else:
                os.environ["UPGPT_LN_FOLD"] = fold
            os.environ["UPGPT_SKIP_FOLD"] = skf
            os.environ["UPGPT_FFOUT_FOLD"] = skf  # (ff.net.2 + proj_out as one GEMM: measured in the same variant)
            unet._plans.clear()
            t0 = time.time()
            pl = unet.plan(B, H, W, ntok, S, "sampler")
            summarize("%s unet sampler B=%d %dx%d ln_fold=%s skip_fold=%s (%.0fs)" % (kind, B, H, W, fold, skf,
                                                                                      time.time() - t0), pl)
            TUNE_CACHE.save(out)
        os.environ.pop("UPGPT_LN_FOLD", None)
        os.environ.pop("UPGPT_SKIP_FOLD", None)
        os.environ.pop("UPGPT_FFOUT_FOLD", None)
        unet._plans.clear()
        if os.environ.get("TUNE_NO_VAE", "0") == "1":
            continue
        t0 = time.time()
        vp = model.first_stage_model._decode_plan(B, H, W, 0.18215)
        summarize("%s vae decode B=%d %dx%d (%.0fs)" % (kind, B, H, W, time.time() - t0), vp)
        TUNE_CACHE.save(out)
    del model
    torch.cuda.empty_cache()
print("entries:", len(TUNE_CACHE.d), "->", out)

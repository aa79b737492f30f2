"""Re-tunes the conv launches of the VAE decoder (and optionally encoder) on the GPU over every configuration, incl. the
big-tile family, with realistic activations in the plan's buffers (a decode of random latents runs first: the MFMA rate
depends on the operand data, all-zero buffers flatter some configurations), and writes the merged tuning cache.

    python scripts/tune_vae.py [out.json] [min_M] [B,h,w ...]     (default shapes: 8,32,32 8,32,24)
"""
import contextlib, io, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd
from upgpt_amd import synth
from upgpt_amd.engine import TUNE_CACHE

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tuned_gfx950.json"
min_m = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
fs = model.first_stage_model


def decode_ms(vp, z, n=10):
    for _ in range(3): vp.run(z)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): vp.run(z)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:]] or [(8, 32, 32), (8, 32, 24)]
for (B, h, w) in shapes:
    z = torch.randn(B, 4, h, w)
    vp = fs._decode_plan(B, h, w, 0.18215)
    before = decode_ms(vp, z)
    ctx = vp.ctx
    changed = 0
    for d, key in vp.convs:
        M = int(key.split("_")[0][1:])
        if M < min_m:
            continue
        old = TUNE_CACHE.get(key)
        cfg, sk, best_us, dflt_us = ctx.conv_autotune(d, 4)
        name = ctx.lib.upk_conv_config_name(cfg).decode()
        if old is None or (int(old[0]), int(old[1])) != (cfg, sk):
            changed += 1
        print("%-52s %s sk %d  %.1f us (was %s)" % (key, name, sk, best_us, old), flush=True)
        TUNE_CACHE.put(key, cfg, sk, best_us, dflt_us)
    TUNE_CACHE.save(out)
    fs._plans.clear()
    vp = fs._decode_plan(B, h, w, 0.18215)
    after = decode_ms(vp, z)
    print("vae decode B=%d %dx%d: %.3f -> %.3f ms (%d launches re-pinned)" % (B, h, w, before, after, changed), flush=True)
# the encoder (img2img / inpainting paths): images of 8 x the latent size
if os.environ.get("TUNE_VAE_ENCODER", "1") == "1":
    for (B, h, w) in shapes:
        x = torch.randn(B, 3, 8 * h, 8 * w).clamp(-1, 1)
        ep = fs._encode_plan(B, 8 * h, 8 * w)

        def enc_ms(ep, n=5):
            for _ in range(2): ep.run(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): ep.run(x)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        before = enc_ms(ep)
        seen = set()
        for d, key in ep.convs:
            M = int(key.split("_")[0][1:])
            if M < min_m or key in seen:
                continue
            seen.add(key)
            old = TUNE_CACHE.get(key)
            cfg, sk, best_us, dflt_us = ep.ctx.conv_autotune(d, 4)
            print("%-52s %s sk %d  %.1f us (was %s)" % (key, ep.ctx.lib.upk_conv_config_name(cfg).decode(), sk, best_us, old), flush=True)
            TUNE_CACHE.put(key, cfg, sk, best_us, dflt_us)
        TUNE_CACHE.save(out)
        fs._enc_plans.clear()
        print("vae encode B=%d %dx%d px: %.3f -> %.3f ms" % (B, 8 * h, 8 * w, before, enc_ms(fs._encode_plan(B, 8 * h, 8 * w))), flush=True)
print("entries:", len(TUNE_CACHE.d), "->", out)

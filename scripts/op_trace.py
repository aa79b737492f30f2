"""Per-op kernel timeline of one UNet forward (dev tool).

  run:    rocprofv3 --kernel-trace --output-format csv -d /tmp/ot -o ot -- python scripts/op_trace.py run [H W]
  parse:  python scripts/op_trace.py parse <kernel_trace.csv> <labels.json>  > table

`run` executes the body program eagerly with a 1-thread marker kernel after every op, so the kernel trace can be
cut into ops; durations are the kernels' own (End - Start), i.e. without launch gaps."""
import contextlib, io, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(H, W, out):
    import torch
    import upgpt_amd
    from upgpt_amd import synth
    with contextlib.redirect_stdout(io.StringIO()):
        model = upgpt_amd.build_model("bbox")
    synth.fill_module_(model); model = model.cuda()
    unet = model.model.diffusion_model
    inp = synth.synth_inputs(8, (H, W), 4, 87, 768, seed=0, text_only=True)
    pl = unet.plan(8, H, W, 87, 50, "sampler")
    pl.load_x_nchw(inp["x_T"].cuda(), 0, 0); pl.load_x_nchw(inp["c_concat"].cuda(), 4, pl.cin_pad)
    pl.load_context(inp["c_crossattn"].cuda()); pl.t_rows.copy_(torch.arange(981, 0, -20, dtype=torch.float32)[:50])
    pl.prep.run()
    ctx = pl.ctx
    mark = torch.zeros(1, dtype=torch.int32, device="cuda")
    s = ctx._s()
    torch.cuda.synchronize()
    for rep in range(4):
        ctx.advance_step(mark); ctx.advance_step(mark); ctx.advance_step(mark)  # triple marker = start of a forward
        # (an op that launches nothing leaves a double one)
        for op in pl.body.ops:
            op(s)
            ctx.advance_step(mark)
        torch.cuda.synchronize()
    json.dump({"labels": pl.body.labels, "cls": pl.body.cls}, open(out, "w"))
    print("ops per forward:", len(pl.body.ops))


def run_vae(H, W, out):
    """Same for the VAE decoder (latent HxW, B=8)."""
    import torch
    import upgpt_amd
    from upgpt_amd import synth
    with contextlib.redirect_stdout(io.StringIO()):
        model = upgpt_amd.build_model("bbox")
    synth.fill_module_(model); model = model.cuda()
    vp = model.first_stage_model._decode_plan(8, H, W, 0.18215)
    vp.z.copy_(torch.randn(8, 4, H, W))
    ctx = vp.ctx
    mark = torch.zeros(1, dtype=torch.int32, device="cuda")
    s = ctx._s()
    torch.cuda.synchronize()
    for rep in range(3):
        ctx.advance_step(mark); ctx.advance_step(mark); ctx.advance_step(mark)
        for op in vp.prog.ops:
            op(s)
            ctx.advance_step(mark)
        torch.cuda.synchronize()
    json.dump({"labels": vp.prog.labels, "cls": vp.prog.cls}, open(out, "w"))
    print("ops per decode:", len(vp.prog.ops))


def parse(trace, labels):
    import csv
    L = json.load(open(labels))
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Grid_Size_X"], r.get("Grid_Size_Z", ""),
           r["Workgroup_Size_X"]) for r in rows]
    is_mark = lambda k: "advance_step" in k[0]
    # forwards start at a double marker
    starts = [i for i in range(len(ks) - 2) if is_mark(ks[i]) and is_mark(ks[i + 1]) and is_mark(ks[i + 2])]
    i0 = starts[-1] + 3
    ops, cur = [], []
    for k in ks[i0:]:
        if is_mark(k):
            ops.append(cur); cur = []
        else:
            cur.append(k)
    assert len(ops) >= len(L["labels"]), (len(ops), len(L["labels"]))
    ops = ops[:len(L["labels"])]
    tot = 0.0
    by_cls = {}
    print("%4s %-9s %-58s %8s  kernels" % ("#", "class", "label", "us"))
    for i, (o, lab, c) in enumerate(zip(ops, L["labels"], L["cls"])):
        us = sum(k[2] - k[1] for k in o) / 1e3
        tot += us
        a = by_cls.setdefault(c, [0, 0.0]); a[0] += 1; a[1] += us
        desc = " + ".join("%s[%s,z%s,%s] %.1f" % (k[0].split("(")[0].replace("void (anonymous namespace)::", "")
                                                   .replace("(anonymous namespace)::", "")[:34], k[3], k[4], k[5], (k[2] - k[1]) / 1e3) for k in o)
        print("%4d %-9s %-58s %8.1f  %s" % (i, c, lab[:58], us, desc))
    print("sum of kernel durations: %.1f us over %d ops" % (tot, len(ops)))
    for c, (n, us) in sorted(by_cls.items()):
        print("  %-10s n=%3d  %8.1f us  (%.1f each)" % (c, n, us, us / n))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
        W = int(sys.argv[3]) if len(sys.argv) > 3 else 32
        (run_vae if os.environ.get("OT_VAE") else run)(H, W, os.environ.get("OT_LABELS", "gpurun_out/ot_labels.json"))
    else:
        parse(sys.argv[2], sys.argv[3])

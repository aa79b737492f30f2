#!/usr/bin/env python
"""bench.py — the reference's headline metric on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1 without a torchrun environment: the script re-launches itself as
   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...,
   one rank per GPU over RCCL; under torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as given)

One "step" = one pass of the hot path over one batch per GPU: x_T -> 50 x (UNetModel forward
+ DDIM update) -> VAE decode -> images on the device (+ one RCCL all-gather of the images
when N > 1).  Workload = BASELINE.json configs[1]: bs=8/GPU, 256x256 px (latent 4x32x32),
50-step DDIM, eta=0, 1 UNet forward per step (no CFG), text-only conditioning [8,87,768],
fp16 kernels with fp32 accumulation, synthetic inputs and recipe weights already resident
in HBM.  value = images/s over all ranks.  The config-true 256x192 (latent 32x24) rate is
reported next to it.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

# multi-process GPU work (RCCL) on this host needs dmabuf IPC; harmless for N = 1
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# the execution lanes want one hardware queue each: 4 is the runtime's default and the most that run side by side
# (profiles/r05_lanes_stream_queue_matrix.txt: with 8, some queue pairs time-slice at 2.5x the serial time)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F16_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_TBS = 8.0             # spec (6.3 TB/s measured for a float4 copy), same guide


def torchrun_command(n, argv, port):
    """The command line `python bench.py --gpus N` becomes: one rank per GPU on this node, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
            "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def relaunch_under_torchrun(n, argv):
    """`python bench.py --gpus N` with no rendezvous in the environment: become
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = torchrun_command(n, argv, port)
    print("[bench] launching %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    os.execv(sys.executable, cmd)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class Workload:
    """bbox.yaml model + synthetic conditioning on one GPU; run() = one bench step."""

    def __init__(self, model, batch, hw, ddim_steps, seed, text_only=True):
        from upgpt_amd import synth
        from upgpt_amd.ddim import DDIMSampler
        self.model, self.B, self.hw, self.S = model, batch, hw, ddim_steps
        inp = synth.synth_inputs(batch, hw, 4, 87, 768, seed=seed, text_only=text_only)
        self.x_T = inp["x_T"].cuda()
        self.cond = {"c_crossattn": inp["c_crossattn"].cuda(), "c_concat": [inp["c_concat"].cuda()]}
        self.sampler = DDIMSampler(model)

    def run(self):
        with self.model.ema_scope():
            z, _ = self.sampler.sample(self.S, self.B, (4,) + tuple(self.hw), self.cond, eta=0.0, x_T=self.x_T,
                                       verbose=False, log_every_t=10 ** 6)
        return self.model.decode_first_stage(z)


def quiet(fn, *a, **k):
    """The sampler prints like the reference does; keep stdout clean for the JSON line."""
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def timed(fn, n, dev):
    """EXACTLY n calls of fn between barrier + device synchronisation on both sides -> (seconds, last result)."""
    from upgpt_amd import dist as D
    sync = (lambda: torch.cuda.synchronize(dev)) if torch.device(dev).type == "cuda" else (lambda: None)
    D.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    sync()
    D.barrier()
    return time.perf_counter() - t0, out


def timed_lanes(pool, step_fn, n, dev, after=None):
    """EXACTLY n bench steps between barrier + device synchronisation on both sides, step k on execution lane
    k % pool.n (upgpt_amd/lanes.py: each lane = own stream, own scratch and buffers, own host thread; the steps of a
    lane run in order, steps of different lanes overlap on the device).  `after(k, images)` runs in step order on
    every rank (the all-gather).  -> (seconds, result of the last step)."""
    from upgpt_amd import dist as D
    sync = (lambda: torch.cuda.synchronize(dev)) if torch.device(dev).type == "cuda" else (lambda: None)
    D.barrier()
    sync()
    t0 = time.perf_counter()
    outs = pool.run(step_fn, n, after=after)
    sync()
    D.barrier()
    return time.perf_counter() - t0, outs[-1]


def gathered(step_fn):
    """One bench step of a rank = its own batch, then the ONE exchange of the path: the all-gather of the decoded
    images (RCCL over xGMI; enqueued on the stream the decode ran on, inside the timed region)."""
    from upgpt_amd import dist as D
    return lambda: D.all_gather_images(step_fn())


def kernel_class_profile(model, wl, reps=20):
    """Per-kernel-class GPU time of one UNet forward, measured live under the SAME conditions
    as the timed region (HIP-graph replay): the forward's launch list is captured once in full
    and once without the class, both replayed `reps` times between HIP events recorded on the
    launch stream; the difference is the class time.  (Bracketing single launches with events
    in eager mode over-reads 10-us kernels by 20-30 %: eager launching is host-bound.)"""
    import ctypes as C
    from upgpt_amd._lib import get_context
    ctx = get_context(torch.cuda.current_device())  # (the rank's own GPU, set by main())
    unet = model.model.diffusion_model
    B, (H, W) = wl.B, wl.hw
    with model.ema_scope():
        plan = unet.plan(B, H, W, 87, wl.S, "sampler")
        body = plan.body
        s = torch.cuda.Stream()

        def timed(skip):
            with torch.cuda.stream(s):
                sp = s.cuda_stream
                ctx._chk(ctx.lib.upk_graph_begin(ctx.h, sp))
                body.run(sp, skip=skip)
                g = C.c_void_p()
                ctx._chk(ctx.lib.upk_graph_end(ctx.h, sp, C.byref(g)))
                ctx._chk(ctx.lib.upk_graph_launch(ctx.h, g, sp))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(reps):
                    ctx._chk(ctx.lib.upk_graph_launch(ctx.h, g, sp))
                e1.record(s)
                s.synchronize()
                ctx.graph_destroy(g)
                return e0.elapsed_time(e1) / reps

        torch.cuda.synchronize()
        timed(())  # (clocks settle)
        names = {"igemm": ("igemm_k1", "igemm_k3"), "attention": ("attention",), "groupnorm": ("groupnorm",),
                 "layernorm": ("layernorm",)}
        out = {}
        fulls = []
        for k, classes in names.items():
            n = sum(1 for c in body.cls if c in classes)
            if n == 0:
                out[k] = {"ms_per_fwd": 0.0, "launches_per_fwd": 0}
                continue
            # full and ablated replays back to back, three times: the median difference is insensitive to the clock
            # drift of a warming GPU (a single full-vs-ablated pair taken minutes apart was off by +-0.09 ms)
            diffs = []
            for _ in range(3):
                f = timed(())
                fulls.append(f)
                diffs.append(f - timed(classes))
            out[k] = {"ms_per_fwd": sorted(diffs)[1], "launches_per_fwd": n}
        full = sorted(fulls)[len(fulls) // 2]
        # GPU kernels per class of one eager pass (a split-K launch = kernel + reduce pass, a GroupNorm with its own
        # statistics = 2): counted by the library (upk_kernel_launches)
        for k, classes in list(names.items()) + [("all", None)]:
            ctx.lib.upk_kernel_launches(ctx.h, 1)
            if classes is None:
                body.run(s.cuda_stream)
            else:
                body.run(s.cuda_stream, skip=tuple(c for c in set(body.cls) if c not in classes))
            n = int(ctx.lib.upk_kernel_launches(ctx.h, 1))
            if classes is None:
                n_kernels = n
            else:
                out[k]["kernels_per_fwd"] = n
        s.synchronize()
        plan.prep.run()  # the ablated replays left garbage in the activations
        torch.cuda.synchronize()
    return out, body.igemm_flops, body.attn_flops, n_kernels, full


def lanes_class_profile(model, pool, wl, reps=8, ncand=6):
    """What a kernel class costs with `pool.n` forwards in flight — the configuration the timed region runs: the captured
    UNet forward of every lane (the lanes' own plans, streams on distinct hardware queues) replayed concurrently, in full and
    without the class in every lane; the wall time per forward and its difference are CHIP time: with the device shared, a
    launch is priced by the CU time it holds, not by its own duration (a profiler's per-kernel durations overlap here)."""
    import ctypes as C
    import upgpt_amd
    unet = model.model.diffusion_model
    plans, streams = [], []
    with model.ema_scope():
        for i in range(pool.n):
            with pool.lane(i):
                plans.append(unet.plan(wl.B, wl.hw[0], wl.hw[1], 87, wl.S, "sampler"))
            streams.append(pool.streams[i] if pool.streams[i] is not None else torch.cuda.current_stream())

        def replay(skip=(), skip_idx=()):
            gs = []
            for p, s in zip(plans, streams):
                ctx = p.ctx
                with torch.cuda.stream(s):
                    ctx._chk(ctx.lib.upk_graph_begin(ctx.h, s.cuda_stream))
                    p.body.run(s.cuda_stream, skip=skip, skip_idx=skip_idx)
                    g = C.c_void_p()
                    ctx._chk(ctx.lib.upk_graph_end(ctx.h, s.cuda_stream, C.byref(g)))
                gs.append(g)
            torch.cuda.synchronize()
            best = None
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    for p, g, s in zip(plans, gs, streams):
                        p.ctx._chk(p.lib.upk_graph_launch(p.hctx, g, s.cuda_stream))
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / (reps * len(plans)) * 1e3
                best = dt if best is None else min(best, dt)
            for p, g in zip(plans, gs):
                p.ctx.graph_destroy(g)
            return best

        replay()
        full = sorted(replay() for _ in range(3))[1]
        out = {}
        for k, classes in (("igemm", ("igemm_k1", "igemm_k3")), ("attention", ("attention",)), ("groupnorm", ("groupnorm",))):
            out[k] = sorted(full - replay(classes) for _ in range(2))[0]
        # the (kernel instantiation, grid) with the largest chip time in this configuration — the lanes' plans carry the
        # launch choices tuned for a shared chip, so the groups are not those of the single-forward figure
        body, ctx = plans[0].body, plans[0].ctx

        def kernel_key(i):
            d, lab = body.meta[i], body.labels[i]
            if d is None or not d.tune_cfg:
                return lab
            up = 2 if d.flags & 0x10 else 1
            M = d.batch * ((d.in_h * up + d.stride - 1) // d.stride) * ((d.in_w * up + d.stride - 1) // d.stride)
            return "%s%s k%d M%d N%d z%d" % (ctx.lib.upk_conv_config_name(d.tune_cfg - 1).decode(), "+app" if (d.c3 or d.c4) else "",
                                            d.ksize, M, d.n_pad, max(1, d.tune_splitk))

        groups = {}
        for i, c in enumerate(body.cls):
            if c.startswith("igemm") or c == "attention":
                groups.setdefault(kernel_key(i), []).append(i)
        exe = lambda i: body.flops[i] * 4.0 / 9.0 if body.labels[i].endswith("_ph") else float(body.flops[i])
        est = lambda idx: sum(6e-6 + exe(i) / 0.4e15 for i in idx)
        best = None
        for key, idx in sorted(groups.items(), key=lambda kv: -est(kv[1]))[:ncand]:
            ms = sorted(full - replay(skip_idx=frozenset(idx)) for _ in range(2))[0]
            if best is None or ms > best[0]:
                best = (ms, key, idx)
        ms, key, idx = best
        fl = sum(exe(i) for i in idx)
        dom = {"label": key, "shapes": sorted(set(body.labels[i] for i in idx)), "launches_per_fwd": len(idx),
               "chip_ms_per_fwd": ms, "chip_us_per_launch": ms * 1e3 / len(idx), "share_of_forward_in_flight": ms / full,
               "method": "as class_ms_per_fwd_in_flight, for one (kernel instantiation, grid) group: the largest of the %d groups "
                         "a static estimate ranks highest" % ncand}
        if fl:
            dom.update({"flops_per_fwd": fl, "achieved": fl / (ms * 1e-3) / 1e12, "frac": fl / (ms * 1e-3) / 1e12 / PEAK_MFMA_F16_TFLOPS})
        for p in plans:
            p.prep.run()  # the ablated replays left garbage in the activations
        torch.cuda.synchronize()
    return full, out, dom


def top_kernel_roofline(model, wl, reps=40):
    """The single conv/GEMM launch with the most executed FLOPs in the forward, replayed back to back under a HIP
    graph (its operands stay L2 / Infinity-Cache warm: an upper bound of what it reaches inside the forward)."""
    import ctypes as C
    unet = model.model.diffusion_model
    with model.ema_scope():
        plan = unet.plan(wl.B, wl.hw[0], wl.hw[1], 87, wl.S, "sampler")
        body, ctx = plan.body, plan.ctx
        # executed FLOPs: a conv behind a nearest-2x upsample runs as four 2x2 phase convs (label "..._ph"), 4/9 of the
        # multiply-adds of the reference's F.interpolate -> conv3x3 that `body.flops` (algorithmic) counts
        exe = lambda f, lab: f * 4.0 / 9.0 if lab.endswith("_ph") else float(f)
        cand = [(exe(f, lab), i) for i, (f, c, lab) in enumerate(zip(body.flops, body.cls, body.labels))
                if c.startswith("igemm") and f]
        if not cand:
            return None
        flops, i = max(cand)
        op = body.ops[i]
        s = torch.cuda.Stream(device=plan.dev)
        with torch.cuda.stream(s):
            sp = s.cuda_stream
            ctx._chk(ctx.lib.upk_graph_begin(ctx.h, sp))
            for _ in range(reps):
                op(sp)
            g = C.c_void_p()
            ctx._chk(ctx.lib.upk_graph_end(ctx.h, sp, C.byref(g)))
            ctx._chk(ctx.lib.upk_graph_launch(ctx.h, g, sp))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(3):
                ctx._chk(ctx.lib.upk_graph_launch(ctx.h, g, sp))
            e1.record(s)
            s.synchronize()
            ctx.graph_destroy(g)
        us = e0.elapsed_time(e1) * 1e3 / (3 * reps)
        plan.prep.run()
        torch.cuda.synchronize()
    tf = flops / (us * 1e-6) / 1e12
    return {"label": body.labels[i], "flops": flops, "flops_kind": "executed by the kernel (the launch with the most of them)",
            "algorithmic_flops": body.flops[i], "us_back_to_back": us, "achieved": tf, "frac": tf / PEAK_MFMA_F16_TFLOPS}


def dominant_kernel_roofline(model, wl, reps=20, ncand=8):
    """The (kernel, grid) with the largest share of the forward's GPU time, timed IN SITU: the launches of the forward
    are grouped by kernel instantiation x grid (what rocprofv3 reports as one row), and for the
    `ncand` groups with the most FLOPs x launches the captured forward is replayed with and without the group (HIP
    events on the launch stream, median of 3 back-to-back pairs); the group with the largest difference is reported
    with its per-launch time inside the forward — cold weights, freshly produced activations, its split-K reduce and
    statistics work included.  (roofline.top_kernel is the other figure: the most-FLOPs launch, warm and isolated.)"""
    import ctypes as C
    unet = model.model.diffusion_model
    with model.ema_scope():
        plan = unet.plan(wl.B, wl.hw[0], wl.hw[1], 87, wl.S, "sampler")
        body, ctx = plan.body, plan.ctx
        # group by what rocprofv3 reports as one row: kernel instantiation (= the tuned tile configuration + loader / epilogue
        # variant a launch descriptor selects) x grid (rows, output columns, split-K) — NOT by shape label: the launches of
        # one instantiation on one grid (e.g. the 3x3 224 -> 224 conv of level 0 with a timestep row, with a residual, with an
        # appended skip segment of 448 or 672 channels) are one kernel to the profiler and to the hardware
        def kernel_key(i):
            d, lab = body.meta[i], body.labels[i]
            if d is None or not d.tune_cfg:
                return lab  # (row-chain / attention launches, untuned convs: the label is the kernel + grid)
            name = ctx.lib.upk_conv_config_name(d.tune_cfg - 1).decode()
            app = "+app" if (d.c3 or d.c4) else ""
            lnf = "+lnf" if d.ln_colsum and not d.ln_rows_in else ""
            M = d.batch * ((d.in_h * (2 if d.flags & 0x10 else 1) + d.stride - 1) // d.stride) * (
                (d.in_w * (2 if d.flags & 0x10 else 1) + d.stride - 1) // d.stride)
            return "%s%s%s k%d M%d N%d z%d" % (name, app, lnf, d.ksize, M, d.n_pad, max(1, d.tune_splitk))

        groups = {}
        for i, (f, c, lab) in enumerate(zip(body.flops, body.cls, body.labels)):
            if c.startswith("igemm") or c == "attention":
                groups.setdefault(kernel_key(i), []).append(i)
        exe = lambda f, lab: f * 4.0 / 9.0 if lab.endswith("_ph") else float(f)
        # candidates: the groups a static estimate (launches x (6 us + FLOPs at 0.4 PFLOP/s)) ranks highest
        est = lambda key, idx: sum(6e-6 + exe(body.flops[i], body.labels[i]) / 0.4e15 for i in idx)
        cands = sorted(groups.items(), key=lambda kv: -est(*kv))[:ncand]
        s = torch.cuda.Stream(device=plan.dev)

        def timed(skip_idx):
            with torch.cuda.stream(s):
                sp = s.cuda_stream
                ctx._chk(ctx.lib.upk_graph_begin(ctx.h, sp))
                body.run(sp, skip_idx=skip_idx)
                g = C.c_void_p()
                ctx._chk(ctx.lib.upk_graph_end(ctx.h, sp, C.byref(g)))
                ctx._chk(ctx.lib.upk_graph_launch(ctx.h, g, sp))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(reps):
                    ctx._chk(ctx.lib.upk_graph_launch(ctx.h, g, sp))
                e1.record(s)
                s.synchronize()
                ctx.graph_destroy(g)
                return e0.elapsed_time(e1) / reps

        torch.cuda.synchronize()
        timed(frozenset())
        best = None
        for lab, idx in cands:
            diffs = []
            for _ in range(3):
                f = timed(frozenset())
                diffs.append(f - timed(frozenset(idx)))
            ms = sorted(diffs)[1]
            if best is None or ms > best[0]:
                best = (ms, lab, idx, f)
        plan.prep.run()  # the ablated replays left garbage in the activations
        torch.cuda.synchronize()
    ms, lab, idx, full = best
    fl = sum(exe(body.flops[i], body.labels[i]) for i in idx)
    out = {"label": lab, "shapes": sorted(set(body.labels[i] for i in idx)), "launches_per_fwd": len(idx), "ms_per_fwd_in_situ": ms, "share_of_forward": ms / full,
           "us_per_launch_in_situ": ms * 1e3 / len(idx), "method": "graph-replay difference with / without the group, "
           "median of 3 pairs; groups = (kernel instantiation, grid) as rocprofv3 rows; candidates: the %d groups a "
           "static estimate ranks highest" % len(cands)}
    if fl:
        tf = fl / (ms * 1e-3) / 1e12
        out.update({"flops_per_fwd": fl, "achieved": tf, "frac": tf / PEAK_MFMA_F16_TFLOPS})
    return out


TRAFFIC_FILE = "profiles/r05_igemm_traffic.json"
PEAK_L2_TBS = 34.5  # aggregate of the eight XCD L2s, /opt/skills/guides/MI355X_MICROARCH.md


def igemm_traffic_bytes_per_launch():
    """(bytes per igemm launch, provenance).  rocprofv3 cannot run inside this process, so the figure is the one the
    PMC passes of scripts/gpu_traffic.sh produced for THIS round's kernels (FETCH_SIZE x2 gfx950 correction +
    WRITE_SIZE, fabric-side) — read from the committed summary and labelled as such; None when it has not been
    collected for the current kernels."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as fh:
            d = json.load(fh)
        src = "%s (%s; collected at commit %s)" % (TRAFFIC_FILE, d.get("command", "PMC"), d.get("commit", "?"))
        # the summary names the kernel sources it was collected for (SHA-256 over csrc/ + upk.h, the hash the build
        # records next to libupk.so): a figure for other kernels is not reported as this build's traffic
        try:
            with open(os.path.join(ROOT, "upgpt_amd", "libupk.so.sha256")) as fh:
                built = fh.read().strip()
        except OSError:
            built = None
        if d.get("kernel_sources_sha256") != built:
            return None, "STALE, not reported: %s was collected for kernel sources %s, this build is %s" % (
                src, str(d.get("kernel_sources_sha256"))[:12], str(built)[:12]), None
        return d["bytes_per_launch"], src, d.get("dominant_kernel_l2")
    except Exception:
        return None, None, None


LANES_TRAFFIC_FILE = "profiles/r06_lanes_traffic.json"   # scripts/lanes_pmc_summary.py over scripts/lanes_replay.py (PMC passes)
LANES_TRACE_FILE = "profiles/r06_lanes_bench_trace.json"  # scripts/trace_frac.py over the kernel trace of this bench command


def _built_sources_hash():
    try:
        with open(os.path.join(ROOT, "upgpt_amd", "libupk.so.sha256")) as fh:
            return fh.read().strip()
    except OSError:
        return None


def committed_summary(rel):
    """A summary a profiler session committed under profiles/ (rocprofv3 cannot run inside this process), or (None, why).
    It is only reported when it names the kernel sources of THIS build (SHA-256 over csrc/ + upk.h, libupk.so.sha256)."""
    try:
        with open(os.path.join(ROOT, rel)) as fh:
            d = json.load(fh)
    except Exception:
        return None, "%s: not collected" % rel
    if d.get("kernel_sources_sha256") != _built_sources_hash():
        return None, "STALE, not reported: %s was collected for kernel sources %s (commit %s), this build is %s" % (
            rel, str(d.get("kernel_sources_sha256"))[:12], d.get("commit", "?"), str(_built_sources_hash())[:12])
    return d, "%s (%s; commit %s)" % (rel, d.get("command", "rocprofv3"), d.get("commit", "?"))


def unet_forward_ms(model, wl, reps=20):
    """Graph-replayed UNet forward + DDIM update (one sampler step), ms."""
    unet = model.model.diffusion_model
    with model.ema_scope():
        plan = unet.plan(wl.B, wl.hw[0], wl.hw[1], 87, wl.S, "sampler")
        st = plan._sampler_state
        plan.step.zero_()
        st.launch(False)
        torch.cuda.synchronize()
        plan.step.zero_()
        t0 = time.perf_counter()
        for _ in range(reps):
            st.launch(False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        plan.step.zero_()
    return dt * 1e3


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(hw, ddim_steps, batch):
    """The CPU oracle (port of the reference algorithm, pinned to the real reference in tests/test_oracle_golden.py)
    timed on this host with SURVEY.md 8(d)'s protocol: CPU model and thread count, 1 warm-up + >= 3 timed UNet
    forwards AT the bench batch (the CPU path is sub-linear in the batch: no scaling from a smaller one), one real
    10-step DDIM loop + VAE decode at B = 1 (BASELINE configs[0]), then images/s extrapolated to the bench workload."""
    from oracle import ddim as o_ddim, schedule as o_sched, unet as o_unet, vae as o_vae
    from upgpt_amd import arch, synth
    cores = min(os.cpu_count() or 1, 32)  # more threads only add oversubscription on this op mix
    torch.set_num_threads(cores)
    shapes = {"model.diffusion_model." + k: v for k, v in arch.UNetArch(**synth.BBOX_UNET).param_shapes().items()}
    shapes.update({"first_stage_model." + k: v for k, v in arch.VAEArch(synth.BBOX_DDCONFIG, 4).param_shapes().items()})
    sd = synth.synth_state_dict(shapes)
    inp = synth.synth_inputs(batch, hw, 4, 87, 768, seed=0, text_only=True)
    x = torch.cat([inp["x_T"], inp["c_concat"]], 1)
    t = torch.full((batch,), 981, dtype=torch.long)
    fwd = lambda n: o_unet.unet_forward(sd, synth.BBOX_UNET, x[:n], t[:n], inp["c_crossattn"][:n])
    t0 = time.perf_counter()
    fwd(batch)  # warm-up at the bench batch
    warm = time.perf_counter() - t0
    times = []
    while len(times) < 3 and (sum(times) < 20 or not times):
        t0 = time.perf_counter()
        fwd(batch)
        times.append(time.perf_counter() - t0)
        if warm > 15 and len(times) >= 1:  # pathologically slow host: keep the run bounded
            break
    t_fwd = min(times)
    # BASELINE configs[0]: single sample, 10-step DDIM + decode, for real
    acp = o_sched.ddpm_tables(o_sched.linear_betas(1000, 0.00085, 0.012))["alphas_cumprod"]
    eps_fn = lambda xx, tt, c: o_unet.diffusion_wrapper(sd, synth.BBOX_UNET, xx, tt, c["c_concat"], c["c_crossattn"])
    cond1 = {"c_crossattn": inp["c_crossattn"][:1], "c_concat": [inp["c_concat"][:1]]}
    t0 = time.perf_counter()
    z, _ = o_ddim.ddim_sample(eps_fn, acp, (1, 4) + tuple(hw), 10, 0.0, inp["x_T"][:1].clone(), cond=cond1)
    t_loop = time.perf_counter() - t0
    t0 = time.perf_counter()
    o_vae.decode_first_stage(sd, synth.BBOX_DDCONFIG, z)
    t_dec = time.perf_counter() - t0
    total = ddim_steps * t_fwd + batch * t_dec
    gf = arch.UNetArch(**synth.BBOX_UNET).flops(batch, hw[0], hw[1], 87) / 1e9
    return {"value": batch / total, "unit": "images/s", "cores": cores, "kind": "port", "cpu": cpu_model_name(),
            "sample": "1 warm-up + %d timed UNet forwards at B=%d latent %dx%d (best %.2f s = %.2f TFLOP/s) and one real "
                      "10-step DDIM loop + VAE decode at B=1 (%.2f s + %.2f s: BASELINE configs[0]); fp32 torch-CPU "
                      "oracle on %d threads; value = %d / (%d x forward + %d x decode)" % (
                          len(times), batch, hw[0], hw[1], t_fwd, gf / t_fwd / 1e3, t_loop, t_dec, cores, batch,
                          ddim_steps, batch),
            "unet_fwd_s": t_fwd, "unet_fwd_times_s": times, "decode_b1_s": t_dec,
            "config0_single_sample_10step_s": t_loop + t_dec}


def encoders_secondary(model, wl, dev, pool=None):
    """BASELINE configs[2] with the conditioning computed on the device: token ids [B, 77] -> CLIP text tower, style
    crops [B, 9, 3, 224, 224] -> CLIP ViT-L/14 image tower, SMPL [B, 1, 85] -> LinearProject, concatenated to the
    [B, 87, 768] context (ddpm.py:734-739), then the main workload (50-step DDIM + decode)."""
    from upgpt_amd import synth
    from upgpt_amd.clip_image import FrozenClipImageEmbedder2
    from upgpt_amd.clip_text import FrozenCLIPEmbedder
    txt = FrozenCLIPEmbedder()
    txt.load_state_dict({k: synth.synth_tensor("cond_stage_model." + k, tuple(v.shape)) for k, v in txt.state_dict().items()})
    img = FrozenClipImageEmbedder2()
    img.load_state_dict({k: synth.synth_tensor("extra_cond_models.0." + k, tuple(v.shape))
                         for k, v in img.state_dict().items()})
    txt, img = txt.cuda(), img.cuda()
    g = torch.Generator(device="cpu").manual_seed(99)
    B = wl.B
    ids = torch.randint(0, 49408, (B, 77), generator=g)
    crops = torch.randn(B, 9, 3, 224, 224, generator=g).cuda()
    smpl = (0.5 * torch.randn(B, 1, 85, generator=g)).cuda()
    pose = model.extra_cond_models[1]

    from upgpt_amd.ddim import DDIMSampler
    n = pool.n if pool is not None else 1
    samplers = [DDIMSampler(model) for _ in range(n)]

    def run(k=0):
        ctx = torch.cat([txt.encode_tokens(ids), img(crops), pose(smpl)], 1)
        with model.ema_scope():
            z, _ = samplers[k % n].sample(wl.S, B, (4,) + tuple(wl.hw), {"c_crossattn": ctx, "c_concat": wl.cond["c_concat"]},
                                          eta=0.0, x_T=wl.x_T, verbose=False, log_every_t=10 ** 6)
        return model.decode_first_stage(z)

    def enc_only():
        return torch.cat([txt.encode_tokens(ids), img(crops), pose(smpl)], 1)

    note = "token ids / pre-processed crops in (tokenizer and crop pre-processing stay on the host)"
    if n > 1:
        with contextlib.redirect_stdout(io.StringIO()):
            timed_lanes(pool, run, n, dev)
            dt, out = timed_lanes(pool, run, 2 * n, dev)
        te, _ = timed(enc_only, 3, dev)
        assert torch.isfinite(out).all()
        return {"value": B * 2 * n / dt, "unit": "images/s", "ms_per_step": dt / (2 * n) * 1e3, "encoders_ms": te / 3 * 1e3,
                "batches_in_flight_per_gpu": n, "note": note}
    quiet(run)
    dt, out = timed(lambda: quiet(run), 2, dev)
    te, _ = timed(enc_only, 3, dev)
    assert torch.isfinite(out).all()
    return {"value": B * 2 / dt, "unit": "images/s", "ms_per_step": dt / 2 * 1e3, "encoders_ms": te / 3 * 1e3, "note": note}


def upscale_secondary(ddim_steps, dev, batch=4, hw=(64, 64), pool=None):
    """BASELINE configs[4] as worded there: the upscale model (models/upgpt/upscale/config.yaml) at bs=4 on a 64x64
    latent, 50-step DDIM; UNet sampling loop only (its kl-f4 first stage is outside the path)."""
    import upgpt_amd
    from upgpt_amd import arch, synth
    from upgpt_amd.ddim import DDIMSampler
    m = quiet(upgpt_amd.build_model, "upscale", overrides={"image_size": list(hw)})
    synth.fill_module_(m)
    m = m.cuda()
    inp = synth.synth_inputs(batch, hw, 3, 86, 768, seed=0, concat_channels=3)
    cond = {"c_crossattn": inp["c_crossattn"].cuda(), "c_concat": [inp["c_concat"].cuda()]}
    x_T = inp["x_T"].cuda()
    n = pool.n if pool is not None else 1
    samplers = [DDIMSampler(m) for _ in range(n)]

    def run(k=0):
        z, _ = samplers[k % n].sample(ddim_steps, batch, (3,) + tuple(hw), cond, eta=0.0, x_T=x_T, verbose=False,
                                      log_every_t=10 ** 6)
        return z

    gf = arch.UNetArch(**synth.UPSCALE_UNET).flops(batch, hw[0], hw[1], 86) / 1e9
    if n > 1:
        with contextlib.redirect_stdout(io.StringIO()):
            timed_lanes(pool, run, n, dev)
            dt, z = timed_lanes(pool, run, 2 * n, dev)
        steps = 2 * n
    else:
        quiet(run)
        dt, z = timed(lambda: quiet(run), 2, dev)
        steps = 2
    assert torch.isfinite(z).all()
    fwd_ms = dt / steps / ddim_steps * 1e3
    return {"value": batch * steps / dt, "unit": "images/s (UNet DDIM loop, no decode)", "ms_per_unet_step": fwd_ms,
            "batches_in_flight_per_gpu": n,
            "algorithmic_gflop_per_fwd": gf, "mfma_util": gf * 1e9 / (fwd_ms * 1e-3) / (PEAK_MFMA_F16_TFLOPS * 1e12)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--latent", default="32x32", help="HxW of the latent (32x32 = 256x256 px)")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("UPGPT_BENCH_LANES", "4")),
                    help="independent bs=8 batches in flight per GPU (execution lanes, upgpt_amd/lanes.py); 1 = one batch "
                         "at a time (the serial loop, also reported as `serial` when lanes > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--cfg", type=float, default=0.0,
                    help="also time the main workload with classifier-free guidance at this scale (UNet on 2*B rows)")
    ap.add_argument("--encoders", action="store_true",
                    help="also time BASELINE configs[2] end to end: CLIP text tower (8 prompts) + CLIP ViT-L/14 image tower "
                         "(8 x 9 style crops) + SMPL projection -> 50-step DDIM -> decode")
    ap.add_argument("--upscale", action="store_true",
                    help="also time BASELINE configs[4]: the upscale UNet, bs=4, 50-step DDIM (UNet loop only), at the 64x64 latent "
                         "BASELINE words and at the 128x96 latent its own config states")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args.gpus, sys.argv[1:])  # (does not return)
    from upgpt_amd import dist as D
    rank, local_rank, world = D.env_world()
    assert world == args.gpus, "WORLD_SIZE %d != --gpus %d" % (world, args.gpus)
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py needs an MI355X per rank: rank %d sees %d GPU(s); there is no CPU fallback" % (
            rank, torch.cuda.device_count() if torch.cuda.is_available() else 0))
    D.init_from_env("nccl" if args.gpus > 1 else None)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import upgpt_amd
    from upgpt_amd import arch, synth
    hw = tuple(int(v) for v in args.latent.split("x"))
    t0 = time.time()
    model = quiet(upgpt_amd.build_model, "bbox")
    synth.fill_module_(model)
    model = model.cuda()
    log("[bench] rank %d model ready in %.1fs" % (rank, time.time() - t0))
    from upgpt_amd.lanes import LanePool
    n_lanes = max(1, min(args.lanes, args.steps))
    # one synthetic batch per lane (lane l of rank r: seed r + 1000 l), all resident in HBM before the timed region
    wls = [Workload(model, args.batch, hw, args.ddim_steps, seed=D.rank_seed(1000 * l, rank)) for l in range(n_lanes)]
    wl = wls[0]
    serial = None
    if n_lanes > 1:
        # the same steps one batch at a time (what `value` was through round 4): lane 0 alone, same barriers, measured
        # BEFORE the lane pool exists, i.e. with the launch choices tuned for one forward having the chip to itself
        ks = max(2, min(args.steps, 6))
        with contextlib.redirect_stdout(io.StringIO()):
            gathered(wl.run)()
            dts, _ = timed(gathered(wl.run), ks, dev)
        dts = D.max_over_ranks(dts, dev)
        serial = {"value": args.batch * world * ks / dts, "unit": "images/s", "steps": ks, "ms_per_step": dts / ks * 1e3,
                  "batches_in_flight_per_gpu": 1}
    pool = LanePool(n_lanes, dev)
    step_k = lambda k: wls[k % n_lanes].run()
    gather_k = lambda k, img: D.all_gather_images(img)

    with contextlib.redirect_stdout(io.StringIO()):  # (the sampler prints like the reference; stdout carries the JSON line)
        # every lane builds its plans / graphs on its first step: warm each lane at least once, W steps in total at least
        # (and twice per lane: the second step of a lane still draws new blocks from the caching allocator, and a
        #  hipMalloc inside the timed region would stall every lane)
        timed_lanes(pool, step_k, max(args.warmup, 2 * n_lanes), dev, after=gather_k)
        dt, out = timed_lanes(pool, step_k, args.steps, dev, after=gather_k)
    lane_ms = pool.lane_step_ms() if n_lanes > 1 else None
    dt = D.max_over_ranks(dt, dev)
    assert out.shape[0] == args.batch * world and torch.isfinite(out).all()
    images = args.batch * world * args.steps
    result = {
        "metric": "256x256 images/sec, 50-step DDIM, bs=8/GPU; UNet MFMA util %",
        "value": images / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: bs=%d/GPU, %dx%d px (latent 4x%dx%d), %d-step DDIM eta=0, "
                               "text-only cond [B,87,768], UNet(bbox.yaml) + VAE decode, EMA weights, no CFG" % (
                                   args.batch, hw[0] * 8, hw[1] * 8, hw[0], hw[1], args.ddim_steps),
                   "batch_per_gpu": args.batch, "ddim_steps": args.ddim_steps, "latent": list(hw),
                   "batches_in_flight_per_gpu": n_lanes, "lane_hw_queues": pool.queue_probe,
                   "lanes": "%d execution lane(s) per GPU: every step is one sample(batch_size=%d) + decode call; step k runs on "
                            "lane k %% %d (own HIP stream, own split-K workspace / activation buffers / step graphs, own host "
                            "thread; packed weights shared), so up to %d independent batches overlap on the device"
                            % (n_lanes, args.batch, n_lanes, n_lanes),
                   "parallelism": "replica-dp%d, one all-gather of images per batch" % world},
    }
    if serial is not None:
        result["serial"] = serial
        # a batch's own time in flight (device time between the ends of consecutive steps of a lane), per lane
        result["step_latency_ms"] = [round(sum(v) / max(1, len(v)), 2) for v in lane_ms]
    if rank == 0:
        a = arch.UNetArch(**synth.BBOX_UNET)
        flops_fwd = a.flops(args.batch, hw[0], hw[1], 87)
        result["mfma_util_whole_job"] = flops_fwd * args.ddim_steps * args.steps / dt / (PEAK_MFMA_F16_TFLOPS * 1e12)
        lanes_prof = lanes_class_profile(model, pool, wl) if n_lanes > 1 else None
        cfg_true_lanes = None
        if n_lanes > 1 and not args.no_secondary and world == 1 and hw != (32, 24):
            wls2 = [Workload(model, args.batch, (32, 24), args.ddim_steps, seed=rank + 1000 * l) for l in range(n_lanes)]
            step2 = lambda k: wls2[k % n_lanes].run()
            k2 = max(n_lanes, (args.steps // 2) // n_lanes * n_lanes)
            with contextlib.redirect_stdout(io.StringIO()):
                timed_lanes(pool, step2, n_lanes, dev)
                dt2l, _ = timed_lanes(pool, step2, k2, dev)
            cfg_true_lanes = {"value": args.batch * k2 / dt2l, "unit": "images/s", "steps": k2, "batches_in_flight_per_gpu": n_lanes}
            del wls2
            # control: the same lanes with TWICE the batch per forward.  Not the configuration BASELINE.json
            # names (bs = 8 per GPU and forward) and never `value`: it shows what the path gives when a caller can batch 16
            wls16 = [Workload(model, 2 * args.batch, hw, args.ddim_steps, seed=rank + 1000 * l + 7) for l in range(n_lanes)]
            step16 = lambda k: wls16[k % n_lanes].run()
            k16 = 2 * n_lanes
            with contextlib.redirect_stdout(io.StringIO()):
                timed_lanes(pool, step16, 2 * n_lanes, dev)
                dt16, _ = timed_lanes(pool, step16, k16, dev)
            result["control_batch16_per_forward"] = {
                "value": 2 * args.batch * k16 / dt16, "unit": "images/s", "batch_per_forward": 2 * args.batch, "steps": k16,
                "batches_in_flight_per_gpu": n_lanes,
                "note": "control, not the headline: %d lanes x bs = %d per sample() call (UNet forward + decode at twice the batch); "
                        "`value` stays at bs = %d per forward" % (n_lanes, 2 * args.batch, args.batch)}
            del wls16
        # everything below describes ONE forward with the chip to itself (the launch choices tuned for that case): kernel
        # quality in isolation, comparable with the earlier rounds' lines
        from upgpt_amd import _lib as _L
        _L.set_concurrency(1)
        fwd_ms = unet_forward_ms(model, wl)
        prof, ig_flops, at_flops, n_kernels, body_ms = kernel_class_profile(model, wl)
        ig_ms = prof["igemm"]["ms_per_fwd"]
        achieved = ig_flops / (ig_ms * 1e-3) / 1e12
        traffic, traffic_src, dom_l2 = igemm_traffic_bytes_per_launch()
        t_model, f_model, b_model = arch.unet_layer_roofline(a, args.batch, hw[0], hw[1], 87, PEAK_MFMA_F16_TFLOPS * 1e12,
                                                             PEAK_HBM_TBS * 1e12)
        n_api, n_k = prof["igemm"]["launches_per_fwd"], prof["igemm"]["kernels_per_fwd"]
        kernel_desc = ("igemm_ws_kernel<*> + igemm_kernel<*> + igemm_as_kernel<*> + igemm_bt_kernel<*> + mlp_kernel<*> + hblock_kernel<*> + "
                       "xblock_kernel<*> + igemm_reduce[_gn|_gnapply]_kernel "
                       "(implicit-GEMM conv / Linear in every tile configuration, the A-stationary Linears, the fused "
                       "row-chain kernels of the 32x32-level transformer blocks (head, cross-attention half, feed-forward "
                       "tail), with the split-K reduce passes and the GroupNorm work they carry)")
        # ---- ONE forward with the chip to itself (the configuration of rounds 1-4; `serial` when lanes are in flight)
        single = {
            "bound": "mfma",
            "kernel": kernel_desc + ": %d conv/GEMM launches = %d kernels per UNet forward" % (n_api, n_k),
            "method": "graph-replay difference: (forward) - (forward without the class), HIP events on the launch stream; "
                      "GroupNorm / LayerNorm work done inside a conv/GEMM launch (split-K reduce pass that normalises, "
                      "folded LayerNorm, statistics by-products) is timed with this class, not with the norm classes",
            "top_kernel": top_kernel_roofline(model, wl),
            "dominant_kernel": dominant_kernel_roofline(model, wl),
            "achieved": achieved, "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_MFMA_F16_TFLOPS,
            "traffic": traffic, "traffic_source": traffic_src,
            # the bound the attribution of round 4 established for the dominant conv is the per-CU L2 -> LDS fill, not the
            # fabric: its L2 request bytes per launch (PMC, committed summary) and the share of the aggregate L2 bandwidth
            # they take over the launch's in-situ duration
            "l2_bytes_per_launch": None, "l2_frac": None,
            "algorithmic_flops_per_fwd": ig_flops, "launches_per_fwd": n_api, "kernels_per_fwd": n_k,
            "avg_launch_us": ig_ms * 1e3 / max(1, n_api), "avg_kernel_us": ig_ms * 1e3 / max(1, n_k),
            # the honest ceiling of the WHOLE forward, per layer max(MFMA time, HBM time)
            "layer_model_ms": t_model * 1e3, "layer_model_bytes_per_fwd": b_model,
            "frac_layer": t_model * 1e3 / fwd_ms,
        }
        dk = single["dominant_kernel"]
        if dom_l2 and dk and dk.get("us_per_launch_in_situ"):
            single["l2_bytes_per_launch"] = dom_l2["bytes_per_launch"]
            single["l2_frac"] = dom_l2["bytes_per_launch"] / (dk["us_per_launch_in_situ"] * 1e-6) / (PEAK_L2_TBS * 1e12)
            single["l2_kernel"] = dom_l2["kernel"]
            single["l2_hit_rate"] = dom_l2["hit_rate"]
        if lanes_prof is None:
            result["roofline"] = single
        else:
            # ---- the TIMED configuration: n_lanes forwards in flight on the shared-chip table.  Every top-level figure of
            # the block describes it; the single-forward figures live under `serial` and nowhere else.
            fwd_l, cls_l, dom_l = lanes_prof
            ach_l = ig_flops / (cls_l["igemm"] * 1e-3) / 1e12
            pmc, pmc_src = committed_summary(LANES_TRAFFIC_FILE)
            trc, trc_src = committed_summary(LANES_TRACE_FILE)
            lanes_launches = (pmc or {}).get("conv_gemm_launches_per_forward") or n_api
            blk = {
                "bound": "mfma",
                "kernel": kernel_desc + ", launch choices of the shared-chip table (tuned_gfx950_lanes.json)",
                "method": "%d forwards in flight (the timed configuration): every lane's captured forward replayed concurrently on "
                          "its own hardware queue, in full and without the class in every lane; (wall time per forward) - (same "
                          "without the class) = the class's chip time per forward (HIP-event / host clock around the replays; a "
                          "profiler's per-kernel durations overlap here).  frac_from_trace recomputes the whole-job figure from the "
                          "committed rocprofv3 kernel trace; traffic / l2_* / mfma_busy are PMC passes over the same replay "
                          "(scripts/lanes_replay.py).  `serial` = ONE forward with the chip to itself, as in rounds 1-4" % n_lanes,
                "achieved": ach_l, "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s", "frac": ach_l / PEAK_MFMA_F16_TFLOPS,
                "forwards_in_flight": n_lanes, "fwd_ms_per_forward_in_flight": fwd_l,
                "class_ms_per_fwd_in_flight": cls_l, "dominant_kernel": dom_l,
                "algorithmic_flops_per_fwd": ig_flops, "launches_per_fwd": lanes_launches,
                "avg_launch_us": cls_l["igemm"] * 1e3 / max(1, lanes_launches),
                "layer_model_ms": t_model * 1e3, "layer_model_bytes_per_fwd": b_model, "frac_layer": t_model * 1e3 / fwd_l,
                # PMC, four forwards in flight: fabric bytes (FETCH_SIZE x2 + WRITE_SIZE) per conv/GEMM launch of a lane-forward
                "traffic": None, "traffic_source": pmc_src,
                "frac_from_trace": None, "frac_from_trace_source": trc_src,
                "serial": single,
            }
            if pmc:
                cls_b = pmc.get("per_lane_forward_conv_gemm_class", {})
                if "FETCH_SIZE" in cls_b:
                    blk["traffic"] = (cls_b["FETCH_SIZE"] + cls_b.get("WRITE_SIZE", 0.0)) / max(1, lanes_launches)
                blk["fabric_bytes_per_lane_forward"] = pmc.get("fabric_bytes_per_lane_forward")
                blk["fabric_bytes_per_forward_one_lane_same_table"] = pmc.get("control_one_lane_same_table_fabric_bytes_per_forward")
                blk["l2_bytes_per_lane_forward"] = (pmc.get("l2_requests_per_lane_forward") or 0) * 128.0 or None
                blk["l2_hit_rate"] = pmc.get("l2_hit_rate")
                blk["mfma_busy_over_gui_active"] = pmc.get("mfma_busy_over_gui_active")
                blk["dominant_kernel_pmc"] = pmc.get("dominant_kernel")
                blk["pmc_dispatch_overlap"] = pmc.get("dispatch_overlap_under_pmc")
            if trc:
                blk["frac_from_trace"] = trc.get("frac_from_trace")
                blk["trace_union_ms_per_forward"] = trc.get("union_ms_per_forward")
                blk["trace_kernels_running_while_busy"] = trc.get("kernels_running_while_busy")
            result["roofline"] = blk
        result["unet"] = {"fwd_ms_graph": fwd_ms, "body_ms_graph": body_ms, "algorithmic_gflop_per_fwd": flops_fwd / 1e9,
                          "mfma_util": flops_fwd / (fwd_ms * 1e-3) / (PEAK_MFMA_F16_TFLOPS * 1e12),
                          "kernel_launches_per_fwd": n_kernels,
                          "class_ms_per_fwd": {k: v["ms_per_fwd"] for k, v in prof.items()},
                          "class_kernels_per_fwd": {k: v["kernels_per_fwd"] for k, v in prof.items()}}
        tdec, _ = timed(lambda: model.decode_first_stage(wl.x_T), 3, dev) if world == 1 else (None, None)
        if tdec is not None:
            result["vae_decode_ms"] = tdec / 3 * 1e3
        if not args.no_secondary and world == 1 and hw != (32, 24):
            wl2 = Workload(model, args.batch, (32, 24), args.ddim_steps, seed=rank)
            quiet(wl2.run)
            dt2, _ = timed(lambda: quiet(wl2.run), max(2, args.steps // 2), dev)
            ser2 = {"value": args.batch * max(2, args.steps // 2) / dt2, "unit": "images/s",
                    "unet_fwd_ms_graph": unet_forward_ms(model, wl2),
                    "algorithmic_gflop_per_fwd": a.flops(args.batch, 32, 24, 87) / 1e9}
            result["config_true_256x192"] = ser2 if cfg_true_lanes is None else dict(cfg_true_lanes, serial=ser2)
        # the other BASELINE configurations, in the bench's own mode (lanes > 1: that many batches in flight)
        sec_pool = pool if n_lanes > 1 else None
        _L.set_concurrency(n_lanes)
        if args.cfg and world == 1:
            uc = {"c_crossattn": torch.zeros_like(wl.cond["c_crossattn"]), "c_concat": wl.cond["c_concat"]}
            from upgpt_amd.ddim import DDIMSampler
            cfg_samplers = [DDIMSampler(model) for _ in range(n_lanes)]

            def run_cfg(k=0):
                with model.ema_scope():
                    z, _ = cfg_samplers[k % n_lanes].sample(wl.S, wl.B, (4,) + tuple(wl.hw), wl.cond, eta=0.0, x_T=wl.x_T,
                                                            verbose=False, log_every_t=10 ** 6,
                                                            unconditional_guidance_scale=args.cfg,
                                                            unconditional_conditioning=uc)
                return model.decode_first_stage(z)

            kc = 2 * n_lanes
            with contextlib.redirect_stdout(io.StringIO()):
                timed_lanes(pool, run_cfg, n_lanes, dev)
                dtc, _ = timed_lanes(pool, run_cfg, kc, dev)
            result["config_cfg"] = {"value": args.batch * kc / dtc, "unit": "images/s", "guidance_scale": args.cfg,
                                    "ms_per_step": dtc / kc * 1e3, "unet_rows": 2 * args.batch,
                                    "batches_in_flight_per_gpu": n_lanes}
        if args.encoders and world == 1:
            result["config_full_cond_with_encoders"] = encoders_secondary(model, wl, dev, sec_pool)
        if args.upscale and world == 1:
            result["config_upscale_bs4_64x64"] = upscale_secondary(args.ddim_steps, dev, pool=sec_pool)
            # the same model at the size its own config states (models/upgpt/upscale/config.yaml:14-16: image_size [128, 96],
            # channels 3): BASELINE configs[4] config-true, 3869.7 GF per forward
            result["config_upscale_config_true"] = dict(upscale_secondary(args.ddim_steps, dev, hw=(128, 96), pool=sec_pool),
                                                        latent="3x128x96", batch=4)
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(hw, args.ddim_steps, args.batch)
        print(json.dumps(result), flush=True)
    D.barrier()


if __name__ == "__main__":
    main()

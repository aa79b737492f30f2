"""In-situ tuning of the UNet forward (dev tool): for every conv/GEMM shape of the plan, tries the candidate (tile
configuration, split-K) pairs INSIDE the captured forward graph (everything else fixed) and keeps the pair with the
lowest replay time — the cold single-launch timings of upk_conv_autotune differ from in-graph behaviour by more than
the gaps between near-tied candidates.  Coordinate descent over the shapes, largest time share first.

    python scripts/tune_insitu.py [H W] [out.json]       (writes a tuning cache with the updated entries)
"""
import contextlib, ctypes as C, io, json, os, sys, time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd
from upgpt_amd import synth
from upgpt_amd.engine import SamplerState, TUNE_CACHE

H = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W = int(sys.argv[2]) if len(sys.argv) > 2 else 32
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/tuned_insitu.json"
REPS = int(os.environ.get("INSITU_REPS", "12"))
TOPK = int(os.environ.get("INSITU_TOPK", "14"))

VAE = os.environ.get("INSITU_VAE", "0") == "1"  # tune the VAE decoder (B = 8, latent HxW) instead of the UNet forward
KIND = os.environ.get("INSITU_KIND", "bbox")     # "upscale": the upscale UNet (3 + 3 channels, 86 tokens)
BATCH = int(os.environ.get("INSITU_B", "8"))     # 16 = the classifier-free-guidance pass over [uncond ; cond]
LANES = int(os.environ.get("INSITU_LANES", "1"))  # > 1: tune for THROUGHPUT with that many forwards in flight (execution
                                                 # lanes, upgpt_amd/lanes.py): the replay time is that of LANES captured
                                                 # forwards replayed concurrently on LANES streams, per forward
if LANES > 1:
    from upgpt_amd import _lib
    from upgpt_amd.tuning import TUNE_CACHE_LANES
    _lib.set_concurrency(LANES)  # (the plans start from the throughput overlay; the output is the updated overlay)
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model(KIND, overrides={"image_size": [H, W]}) if KIND == "upscale" else upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
unet = model.model.diffusion_model
NTOK, CH = (86, 3) if KIND == "upscale" else (87, 4)
inp = (synth.synth_inputs(BATCH, (H, W), 3, 86, 768, seed=0, concat_channels=3) if KIND == "upscale"
       else synth.synth_inputs(BATCH, (H, W), 4, 87, 768, seed=0, text_only=True))
if VAE:
    vp = model.first_stage_model._decode_plan(8, H, W, 0.18215)
    vp.z.copy_(torch.randn(8, 4, H, W))
pl = unet.plan(BATCH, H, W, NTOK, 50, "sampler")
pl.load_x_nchw(inp["x_T"].cuda(), 0, 0); pl.load_x_nchw(inp["c_concat"].cuda(), CH, pl.cin_pad)
pl.load_context(inp["c_crossattn"].cuda()); pl.t_rows.copy_(torch.arange(981, 0, -20, dtype=torch.float32)[:50])
pl.prep.run()
st = SamplerState(pl, CH); st.x.copy_(inp["x_T"].cuda()); st.coefs.fill_(0.5)
ctx = pl.ctx
lane_plans, lane_states, lane_streams = [pl], [st], [torch.cuda.current_stream()]
if LANES > 1:
    from upgpt_amd.lanes import LanePool
    _pool = LanePool(LANES)  # (its streams are probed to sit on distinct hardware queues)
    print("lane streams on distinct hardware queues:", _pool.queue_probe, flush=True)
    lane_streams = list(_pool.streams)
for i in range(1, LANES):
    with upgpt_amd.lane(i):
        p2 = unet.plan(BATCH, H, W, NTOK, 50, "sampler")
        p2.load_x_nchw(inp["x_T"].cuda(), 0, 0); p2.load_x_nchw(inp["c_concat"].cuda(), CH, p2.cin_pad)
        p2.load_context(inp["c_crossattn"].cuda()); p2.t_rows.copy_(torch.arange(981, 0, -20, dtype=torch.float32)[:50])
        p2.prep.run()
        s2 = SamplerState(p2, CH); s2.x.copy_(inp["x_T"].cuda()); s2.coefs.fill_(0.5)
    lane_plans.append(p2); lane_states.append(s2)
torch.cuda.synchronize()
ncfg = ctx.lib.upk_conv_num_configs()
names = [ctx.lib.upk_conv_config_name(i).decode() for i in range(ncfg)]
ONLY = os.environ.get("INSITU_ONLY", "")
KEYSUB = os.environ.get("INSITU_KEYS", "")  # only the shapes whose key contains this substring
ONLY3 = os.environ.get("INSITU_ONLY3", "")  # only the 3x3 shapes, only the configurations whose names start with this prefix


def replay_ms():
    if VAE:
        vp.prog.run(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(max(2, REPS // 3)): vp.prog.run()
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / max(2, REPS // 3) * 1e3)
        return best
    if LANES > 1:
        return replay_lanes_ms()
    for g in list(st.graphs.values()):
        ctx.graph_destroy(g)
    st.graphs.clear()
    st.launch(False); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        pl.step.zero_(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(REPS): st.launch(False)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / REPS * 1e3)
    return best


def replay_lanes_ms():
    """LANES forwards in flight: every lane's graph re-captured with the current descriptors, then REPS replays per lane
    enqueued round robin on the lanes' streams; wall time per forward."""
    gs = []
    for i, (p, s_, strm) in enumerate(zip(lane_plans, lane_states, lane_streams)):
        for g in list(s_.graphs.values()):
            p.ctx.graph_destroy(g)
        s_.graphs.clear()
        with upgpt_amd.lane(i, strm):
            gs.append(s_.graph(False))
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        for p in lane_plans:
            p.step.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(REPS):
            for p, g, strm in zip(lane_plans, gs, lane_streams):
                p.ctx._chk(p.lib.upk_graph_launch(p.hctx, g, strm.cuda_stream))
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / (REPS * LANES) * 1e3)
    return best


# body convs only (the plan's conv list also holds the step-invariant prep launches)
groups = {}
if VAE:
    body = set(id(k) for k in vp.prog.keep)
    for d, key in vp.convs:
        if id(d) in body:
            groups.setdefault(key, []).append(d)
else:
    for p in lane_plans:  # (the same shape is pinned to the same choice in every lane)
        body = set(id(k) for k in p.body.keep)
        for d, key in p.convs:
            if id(d) in body:
                groups.setdefault(key, []).append(d)
NOISE = 0.004 if VAE else (0.006 if LANES > 1 else 0.0015)  # (concurrent replays are noisier)
print("shapes in the forward:", len(groups), "launch descriptors:", sum(len(v) for v in groups.values()), flush=True)
base = replay_ms()
print("baseline forward %.4f ms" % base, flush=True)


def feasible(d, cfg, sk):
    old = (d.tune_cfg, d.tune_splitk)
    d.tune_cfg, d.tune_splitk = cfg + 1, sk
    m, n = C.c_int(0), C.c_int(0)
    rc = ctx.lib.upk_conv_gn_fused(ctx.h, C.byref(d), C.byref(m), C.byref(n))
    d.tune_cfg, d.tune_splitk = old
    return rc == 0


# order: tuned time x launches, largest first
def share(key):
    e = TUNE_CACHE.get(key) or TUNE_CACHE.get(key[:-3] if key.endswith("_gs") else key)
    return (e[2] if e else 10.0) * len(groups[key])


START_CHIP = LANES > 1 and os.environ.get("INSITU_START_CHIP", "0") == "1"
# the choices the run starts with (what the written table is compared against)
ORIG = {key: ((groups[key][0].tune_cfg - 1, groups[key][0].tune_splitk) if groups[key][0].tune_cfg > 0 else None) for key in groups}
CHIP_RANK = {}  # key -> candidates ranked by chip time (START_CHIP)


def chip_us(ds_lane, cfg, sk, R=6):
    """CHIP time per launch of one shape in configuration (cfg, sk): R back-to-back launches captured per lane, the LANES
    graphs replayed concurrently — what the launch costs when four chains of ITSELF share the chip (a launch that fills
    every CU with one workgroup pays its full latency, one with fewer / lighter workgroups packs)."""
    olds = [(d.tune_cfg, d.tune_splitk) for d in ds_lane]
    gs = []
    try:
        for d in ds_lane:
            d.tune_cfg, d.tune_splitk = cfg + 1, sk
        for p, d, strm in zip(lane_plans, ds_lane, lane_streams):
            with torch.cuda.stream(strm):
                p.ctx._chk(p.lib.upk_graph_begin(p.hctx, strm.cuda_stream))
                try:
                    for _ in range(R):
                        p.ctx.conv(d)
                finally:
                    g = C.c_void_p(); rc = p.lib.upk_graph_end(p.hctx, strm.cuda_stream, C.byref(g))
                p.ctx._chk(rc)
            gs.append(g)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                for p, g, strm in zip(lane_plans, gs, lane_streams):
                    p.ctx._chk(p.lib.upk_graph_launch(p.hctx, g, strm.cuda_stream))
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / (3 * R * len(gs)) * 1e6)
        return best
    finally:
        for p, g in zip(lane_plans, gs):
            p.ctx.graph_destroy(g)
        for d, o in zip(ds_lane, olds):
            d.tune_cfg, d.tune_splitk = o


if START_CHIP:
    # phase A: every shape to the configuration with the lowest chip time of its own (all at once: a coordinate descent
    # from the latency-tuned table cannot leave it one shape at a time — a launch with few large tiles gains nothing while
    # every other lane still fills the CUs with one-workgroup-per-CU launches)
    moved = 0
    for key in sorted(groups, key=share, reverse=True):
        if KEYSUB and KEYSUB not in key:
            continue
        ds = groups[key]
        per = len(ds) // LANES
        ds_lane = [ds[i * per] for i in range(LANES)]
        d0 = ds[0]
        start = (d0.tune_cfg - 1, d0.tune_splitk) if d0.tune_cfg > 0 else None
        sks = sorted({1, 2, 3, 4, 6, 8} | ({start[1]} if start else set()))
        rank = []
        for c in range(ncfg):
            if names[c].startswith("as") and d0.ksize != 1:
                continue
            for s_ in sks:
                if not feasible(d0, c, s_):
                    continue
                try:
                    rank.append((chip_us(ds_lane, c, s_), (c, s_)))
                except Exception:
                    torch.cuda.synchronize()
        rank.sort()
        CHIP_RANK[key] = [cs for _, cs in rank[:TOPK]]
        if rank and rank[0][1] != start:
            t_start = [t for t, cs in rank if cs == start]
            print("chip  %-46s %s %.2f us -> %s %.2f us" % (key, (names[start[0]], start[1]) if start else None,
                                                           t_start[0] if t_start else float("nan"),
                                                           (names[rank[0][1][0]], rank[0][1][1]), rank[0][0]), flush=True)
            for d in ds:
                d.tune_cfg, d.tune_splitk = rank[0][1][0] + 1, rank[0][1][1]
            moved += 1
    t_all = replay_ms()
    print("phase A: %d shapes moved to their chip-time best: forward %.4f -> %.4f ms" % (moved, base, t_all), flush=True)

changed = {}
cur = replay_ms() if START_CHIP else base
MAXSHAPES = int(os.environ.get("INSITU_MAXSHAPES", "0"))  # only the N shapes with the largest time share
for rank_, key in enumerate(sorted(groups, key=share, reverse=True)):
    if KEYSUB and KEYSUB not in key:
        continue
    if MAXSHAPES and rank_ >= MAXSHAPES:
        break
    ds = groups[key]
    d0 = ds[0]
    start = (d0.tune_cfg - 1, d0.tune_splitk) if d0.tune_cfg > 0 else None
    sks = sorted({1, 2, 3, 4, 6, 8, 9} | ({start[1]} if start else set()))
    if ONLY3:
        if d0.ksize != 3:
            continue
        cands = [(c, s) for c in range(ncfg) if names[c].startswith(ONLY3) for s in (1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 14)
                 if feasible(d0, c, s)]
        if not cands:
            continue
    elif ONLY:  # e.g. INSITU_ONLY=as: only the A-stationary family's configurations (second slot = passes per workgroup)
        if d0.ksize != 1:
            continue
        cands = [(c, s) for c in range(ncfg) if names[c].startswith(ONLY) for s in (1, 2, 3, 4, 6, 8, 12, 16)
                 if feasible(d0, c, s)]
        if not cands:
            continue
    elif START_CHIP and key in CHIP_RANK:
        cands = list(CHIP_RANK[key])
        for extra in (start, ORIG.get(key)):  # (the phase-A choice and the table's own: the descent may go back)
            if extra and extra not in cands:
                cands.append(extra)
    else:
        cands = [(c, s) for c in range(ncfg) for s in sks if feasible(d0, c, s)]
    # keep the candidates the single-launch tuner ranks near the top (plus the current choice)
    timed = []
    if len(cands) > TOPK:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for (c, s) in cands:
            d0.tune_cfg, d0.tune_splitk = c + 1, s
            try:
                ctx.conv(d0)
                ev0.record()
                for _ in range(4): ctx.conv(d0)
                ev1.record(); torch.cuda.synchronize()
                timed.append((ev0.elapsed_time(ev1), (c, s)))
            except Exception:
                pass
        timed.sort()
        cands = [cs for _, cs in timed[:TOPK]]
        if start and start not in cands:
            cands.append(start)
    best_cs, best_t = start, cur
    for (c, s) in cands:
        if (c, s) == start:
            continue
        for d in ds:
            d.tune_cfg, d.tune_splitk = c + 1, s
        try:
            t = replay_ms()
        except Exception as e:
            t = 1e9
        if t < best_t - NOISE:  # above the replay noise
            best_cs, best_t = (c, s), t
    for d in ds:
        d.tune_cfg, d.tune_splitk = (best_cs[0] + 1, best_cs[1]) if best_cs else (0, 0)
    if best_cs != start:
        # confirm against the previous choice once more
        t_new = replay_ms()
        for d in ds:
            d.tune_cfg, d.tune_splitk = (start[0] + 1, start[1]) if start else (0, 0)
        t_old = replay_ms()
        if t_new < t_old - NOISE * 0.7:
            for d in ds:
                d.tune_cfg, d.tune_splitk = best_cs[0] + 1, best_cs[1]
            changed[key] = (start, best_cs, t_old, t_new)
            cur = t_new
            print("%-50s x%d  %s -> %s   %.4f -> %.4f ms" % (key, len(ds), start, best_cs, t_old, t_new), flush=True)
        else:
            cur = t_old
final = replay_ms()
print("forward %.4f -> %.4f ms, %d shapes changed" % (base, final, len(changed)))
ent = dict(TUNE_CACHE_LANES.d) if LANES > 1 else dict(TUNE_CACHE.d)
for key, (start, best, t_old, t_new) in changed.items():
    old = ent.get(key) or ent.get(key[:-3]) or TUNE_CACHE.get(key) or [0, 1, 0.0, 0.0]
    ent[key] = [best[0], best[1], old[2], old[3]]
if START_CHIP:  # every shape whose final choice differs from the one the run started with (phase A + descent)
    for key, ds in groups.items():
        fin = (ds[0].tune_cfg - 1, ds[0].tune_splitk) if ds[0].tune_cfg > 0 else None
        if fin and fin != ORIG.get(key):
            old = ent.get(key) or TUNE_CACHE.get(key) or [0, 1, 0.0, 0.0]
            ent[key] = [fin[0], fin[1], old[2], old[3]]
if LANES > 1:
    ent.update(TUNE_CACHE_LANES.meta)
ent["__configs__"] = names  # (the indices refer to THIS library's configuration list: TuneCache.bind)
json.dump(ent, open(out, "w"), indent=0, sort_keys=True)
json.dump({k: [list(v[0]) if v[0] else None, list(v[1]), v[2], v[3]] for k, v in changed.items()},
          open(out.replace(".json", "_changes.json"), "w"), indent=0, sort_keys=True)
print("wrote", out)

"""Round-6 lane experiments on one MI355X.  Every experiment replays the captured UNet forward
(B x 32 x 32 latents, bbox.yaml UNet, 87 context tokens) of several lanes concurrently and reports WALL time per forward =
chip time per lane-forward, the quantity the bench's images/s follows.

  r6_lanes_lab.py cumask     lanes on disjoint CU sets (hipExtStreamCreateWithCUMask through upk_stream_create_cumask):
                             4 x 2 XCDs, 2 x 4 XCDs, 8 x 1 XCD against the shared chip, with the latency table and with the
                             shared-chip table; checks that captured graphs honour the mask and that the masked streams sit on
                             distinct hardware queues
  r6_lanes_lab.py align      do four lanes that run the SAME layer at the same time share its weight fetch?  free-running
                             (today), re-aligned at every forward (one cross-stream event barrier per forward), and staggered
                             by a quarter forward
  r6_lanes_lab.py controls   (lanes, B) in {(1,8), (1,16), (1,32), (2,16), (4,8), (4,16)}: ms per forward and per 8 samples
"""
import contextlib, ctypes as C, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd
from upgpt_amd import _lib as L
from upgpt_amd import synth
from upgpt_amd.lanes import LanePool, cu_partition_streams

MODE = sys.argv[1] if len(sys.argv) > 1 else "cumask"
REPS = int(os.environ.get("LAB_REPS", "8"))
H = W = 32

with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model)
model = model.cuda()
unet = model.model.diffusion_model


_inputs = {}


def lane_plan(i, stream, conc, B=8):
    """Lane i's sampler plan with real (synthetic, seeded) operands in its buffers: all-zero activations run 12-22 % faster
    than real ones on this chip (switching power), so timing empty buffers would flatter every variant."""
    if B not in _inputs:
        inp = synth.synth_inputs(B, (H, W), 4, 87, 768, seed=0, text_only=True)
        _inputs[B] = {k: inp[k].cuda() for k in ("x_T", "c_concat", "c_crossattn")}
    inp = _inputs[B]
    with L.lane(i, stream, concurrency=conc):
        p = unet.plan(B, H, W, 87, 50, "sampler")
        p.load_x_nchw(inp["x_T"], 0, 0)
        p.load_x_nchw(inp["c_concat"], 4, p.cin_pad)
        p.load_context(inp["c_crossattn"])
        p.t_rows.copy_(torch.arange(981, 0, -20, dtype=torch.float32)[:50])
        p._t_rows_key = None
        p.prep.run()
    return p


def capture(p, s):
    ctx = p.ctx
    with torch.cuda.stream(s):
        ctx._chk(ctx.lib.upk_graph_begin(ctx.h, s.cuda_stream))
        p.body.run(s.cuda_stream)
        g = C.c_void_p()
        ctx._chk(ctx.lib.upk_graph_end(ctx.h, s.cuda_stream, C.byref(g)))
    return g


def replay(plans, streams, reps=REPS, before_each=None, tries=3):
    """ms per forward (wall / (reps * lanes)), best of `tries`."""
    gs = [capture(p, s) for p, s in zip(plans, streams)]
    torch.cuda.synchronize()
    for p, g, s in zip(plans, gs, streams):  # warm
        p.ctx._chk(p.lib.upk_graph_launch(p.hctx, g, s.cuda_stream))
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(tries):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(reps):
            if before_each is not None:
                before_each(r)
            for p, g, s in zip(plans, gs, streams):
                p.ctx._chk(p.lib.upk_graph_launch(p.hctx, g, s.cuda_stream))
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (reps * len(plans)) * 1e3)
    for p, g in zip(plans, gs):
        p.ctx.graph_destroy(g)
    return best


def graph_probe(ctx, s, nblocks=1024):
    """XCC ids seen by the probe kernel when it runs as a node of a captured graph launched on stream s."""
    out = torch.zeros(nblocks * 2, dtype=torch.int32, device="cuda")
    with torch.cuda.stream(s):
        ctx._chk(ctx.lib.upk_graph_begin(ctx.h, s.cuda_stream))
        ctx._chk(ctx.lib.upk_probe_placement(ctx.h, out.data_ptr(), nblocks, 40000, C.c_void_p(s.cuda_stream)))
        g = C.c_void_p()
        ctx._chk(ctx.lib.upk_graph_end(ctx.h, s.cuda_stream, C.byref(g)))
    ctx._chk(ctx.lib.upk_graph_launch(ctx.h, g, s.cuda_stream))
    torch.cuda.synchronize()
    ctx.graph_destroy(g)
    v = out.cpu().view(nblocks, 2)
    seen = {}
    for x, cu in zip((v[:, 0] & 0xF).tolist(), ((v[:, 1] >> 8) & 0xFF).tolist()):
        seen.setdefault(x, set()).add(cu)
    return {x: len(c) for x, c in sorted(seen.items())}


if MODE == "cumask":
    pool = LanePool(4)
    ctx0 = L.get_context(0, lane=0)
    print("shared-chip lane streams on distinct hardware queues:", pool.queue_probe, flush=True)
    rows = []

    def run(label, streams, conc):
        plans = [lane_plan(i, s, conc) for i, s in enumerate(streams)]
        torch.cuda.synchronize()
        alone = replay(plans[:1], streams[:1])
        allms = replay(plans, streams)
        rows.append((label, len(streams), alone, allms))
        print("%-58s lanes %d: one lane alone %.3f ms per forward, all lanes %.3f ms per forward" % (label, len(streams), alone, allms), flush=True)

    run("shared chip (256 CUs), latency table", list(pool.streams), 1)
    run("shared chip (256 CUs), shared-chip table", list(pool.streams), 4)
    for n in (4, 2, 8):
        tag = "%d x %d CUs (CU slots, every XCD)" % (n, 256 // n)
        try:
            streams, seen = cu_partition_streams(ctx0, n)
        except Exception as e:
            print("%s: %s" % (tag, e), flush=True)
            continue
        print("%s: probe placement per stream %s" % (tag, seen), flush=True)
        print("  inside a captured graph the probe of stream 0 saw XCDs:", graph_probe(ctx0, streams[0]), flush=True)
        ov = [pool._overlap(streams[0], s, 200000)[0] for s in streams[1:]]
        print("  stream 0 overlaps with the others (distinct hardware queues):", ov, flush=True)
        run(tag + ", latency table", streams, 1)
        run(tag + ", shared-chip table", streams, 4)
    print("\n%-58s %5s %12s %12s %10s" % ("configuration", "lanes", "alone ms", "in flight ms", "img/s UNet"))
    for label, n, a, b in rows:
        print("%-58s %5d %12.3f %12.3f %10.1f" % (label, n, a, b, 8 / (b * 50) * 1e3))

elif MODE == "maskdiag":
    # what does a CU mask do on this box?  For a few masks: XCDs and CUs the probe kernel's workgroups land on
    import collections
    ctx0 = L.get_context(0, lane=0)

    def words(bits):
        w = [0] * 8
        for b in bits:
            w[b // 32] |= 1 << (b % 32)
        return w

    def hist(stream, nblocks=4096, spin=60000):
        out = torch.zeros(nblocks * 2, dtype=torch.int32, device="cuda")
        ctx0._chk(ctx0.lib.upk_probe_placement(ctx0.h, out.data_ptr(), nblocks, spin, C.c_void_p(stream.cuda_stream)))
        torch.cuda.synchronize()
        v = out.cpu().view(nblocks, 2)
        h = collections.defaultdict(set)
        for x, hw in zip((v[:, 0] & 0xF).tolist(), ((v[:, 1] >> 8) & 0xFF).tolist()):
            h[x].add(hw)
        return {x: len(c) for x, c in sorted(h.items())}

    print("unmasked torch stream:", hist(torch.cuda.Stream()), flush=True)
    cases = [("bit 0", [0]), ("bits 0-7", range(8)), ("bits 0-31", range(32)), ("bits 0-63", range(64)), ("bits 0-127", range(128)),
             ("bits 128-255", range(128, 256)), ("every 8th bit (b % 8 == 0)", range(0, 256, 8)), ("b % 8 in {0,1}", [b for b in range(256) if b % 8 < 2]),
             ("b % 16 < 2", [b for b in range(256) if b % 16 < 2]), ("all 256", range(256)), ("bits 32-63", range(32, 64)), ("bits 224-255", range(224, 256))]
    for nwords in (8, 10):
        for name, bits in cases:
            w = words(bits) + [0] * (nwords - 8)
            arr = (C.c_uint32 * len(w))(*w)
            h = C.c_void_p()
            rc = ctx0.lib.upk_stream_create_cumask(ctx0.h, arr, len(w), C.byref(h))
            if rc != 0:
                print("%d words, %-28s: create failed: %s" % (nwords, name, (ctx0.lib.upk_last_error(ctx0.h) or b"").decode()), flush=True)
                continue
            s = torch.cuda.ExternalStream(h.value)
            hh = hist(s)
            print("%d words, %-28s: %d CUs on XCDs %s" % (nwords, name, sum(hh.values()), hh), flush=True)
            ctx0.lib.upk_stream_destroy(ctx0.h, C.c_void_p(s.cuda_stream))

elif MODE == "morelanes":
    # more than four batches in flight: the runtime multiplexes ordinary streams onto 4 hardware queues, a stream created with a
    # CU mask gets a queue of its own — with ALL CUs allowed it is an ordinary lane stream on a fifth, sixth, ... queue
    import ctypes
    pool = LanePool(4)
    ctx0 = L.get_context(0, lane=0)
    full = (ctypes.c_uint32 * 8)(*([0xFFFFFFFF] * 8))
    xs = []
    for _ in range(8):
        h = ctypes.c_void_p()
        ctx0._chk(ctx0.lib.upk_stream_create_cumask(ctx0.h, full, 8, ctypes.byref(h)))
        xs.append(torch.cuda.ExternalStream(h.value))
    ov = [[int(pool._overlap(a, b, 200000)[0]) if a is not b else 0 for b in xs] for a in xs]
    print("pairwise overlap of 8 full-mask streams (1 = run concurrently):")
    for r in ov:
        print("   ", r)
    ov2 = [int(pool._overlap(pool.streams[0], b, 200000)[0]) for b in xs]
    print("pool stream 0 against them:", ov2, flush=True)
    base = [lane_plan(i, s, 4) for i, s in enumerate(pool.streams)]
    print("4 lanes, runtime streams (bench mode)      : %.3f ms per forward" % replay(base, list(pool.streams)), flush=True)
    for n in (4, 5, 6, 8):
        plans = [lane_plan(i, s, 4) for i, s in enumerate(xs[:n])]
        print("%d lanes, full-mask streams (own queues)    : %.3f ms per forward" % (n, replay(plans, xs[:n])), flush=True)

elif MODE in ("knobs", "prefetch"):
    # row-chain kernels at the 16x16 level with the chip shared: 64-workgroup launches are what four lanes want
    from upgpt_amd import knobs
    pool = LanePool(4)
    streams = list(pool.streams)
    SETS = (("default", {}), ("XBLOCK=1", {"XBLOCK": "1"}), ("MLP_FUSE=1", {"MLP_FUSE": "1"}), ("XBLOCK=1 MLP_FUSE=1", {"XBLOCK": "1", "MLP_FUSE": "1"}),
            ("XBLOCK=1 XB_ROWS=16", {"XBLOCK": "1", "XB_ROWS": 16}), ("default again", {}))
    if MODE == "prefetch":  # next-weight prefetch (include/upk.h pf_next) off / on, and how much of the next weight
        SETS = (("WEIGHT_PREFETCH=0", {"WEIGHT_PREFETCH": "0"}), ("WEIGHT_PREFETCH=1", {"WEIGHT_PREFETCH": "1"}),
                ("WEIGHT_PREFETCH=0", {"WEIGHT_PREFETCH": "0"}), ("WEIGHT_PREFETCH=1", {"WEIGHT_PREFETCH": "1"}),
                ("WEIGHT_PREFETCH=1 AHEAD=2", {"WEIGHT_PREFETCH": "1", "WEIGHT_PREFETCH_AHEAD": 2}),
                ("WEIGHT_PREFETCH=1 AHEAD=3", {"WEIGHT_PREFETCH": "1", "WEIGHT_PREFETCH_AHEAD": 3}))
    for name, kv in SETS:
        old = {k: getattr(knobs, k) for k in kv}
        for k, v in kv.items():
            setattr(knobs, k, v)
        for pl in list(unet._plans.values()):
            pl.close()
        unet._plans.clear()
        try:
            plans = [lane_plan(i, s, 4) for i, s in enumerate(streams)]
            labs = [l for l in plans[0].body.labels if l.startswith(("xblock", "mlp", "hblock"))]
            ms = replay(plans, streams, reps=12)
            if MODE == "prefetch":
                one = replay(plans[:1], streams[:1], reps=12)
                for pl in list(unet._plans.values()):
                    pl.close()
                unet._plans.clear()
                ser = lane_plan(0, streams[0], 1)  # the latency table: one forward with the chip to itself (`serial`)
                lat = replay([ser], streams[:1], reps=12)
                print("%-28s %.3f ms per forward in flight, %.3f ms one lane alone; latency table alone %.3f ms (%d prefetch links)" % (
                    name, ms, one, lat, ser.n_prefetch_links), flush=True)
            else:
                print("%-24s %.3f ms per forward (%d kernels; row-chain ops: %s)" % (name, ms, len(plans[0].body.labels), sorted(set(labs))), flush=True)
        except Exception as e:
            print("%-24s failed: %s" % (name, str(e)[:200]), flush=True)
        for k, v in old.items():
            setattr(knobs, k, v)

elif MODE == "fwd":
    # the three forward times of the current environment (A/B of library-level switches, one process per setting)
    pool = LanePool(4)
    streams = list(pool.streams)
    plans = [lane_plan(i, s, 4) for i, s in enumerate(streams)]
    ms = replay(plans, streams, reps=12)
    one = replay(plans[:1], streams[:1], reps=12)
    ser = lane_plan(0, streams[0], 1)
    lat = replay([ser], streams[:1], reps=12)
    print("%s: %.3f ms per forward in flight, %.3f ms one lane alone; latency table alone %.3f ms" % (
        os.environ.get("LAB_TAG", ""), ms, one, lat), flush=True)

elif MODE == "clock":
    # shader clock under load: a spinning probe wave on a fifth stream while 0 / 1 / 4 lanes replay forwards
    pool = LanePool(4)
    streams = list(pool.streams)
    plans = [lane_plan(i, s, 4) for i, s in enumerate(streams)]
    gs = [capture(p, s) for p, s in zip(plans, streams)]
    ctx0 = L.get_context(0, lane=0)
    ps = torch.cuda.Stream()
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    for nl in (0, 1, 4, 4):
        torch.cuda.synchronize()
        for r in range(40 if nl else 0):  # ~60 ms of lanes work per probe
            for p, g, s in list(zip(plans, gs, streams))[:nl]:
                p.ctx._chk(p.lib.upk_graph_launch(p.hctx, g, s.cuda_stream))
            if r == 8:
                ctx0._chk(ctx0.lib.upk_probe_clock(ctx0.h, out.data_ptr(), 2_000_000, C.c_void_p(ps.cuda_stream)))  # 20 ms
        if not nl:
            ctx0._chk(ctx0.lib.upk_probe_clock(ctx0.h, out.data_ptr(), 2_000_000, C.c_void_p(ps.cuda_stream)))
        torch.cuda.synchronize()
        c, w = out.tolist()
        print("%d lanes replaying: shader clock %.0f MHz over %.1f ms (cycles %d, 100 MHz ticks %d)" % (nl, 100.0 * c / max(1, w), w / 1e5, c, w), flush=True)

elif MODE == "align":
    pool = LanePool(4)
    streams = list(pool.streams)
    plans = [lane_plan(i, s, 4) for i, s in enumerate(streams)]
    torch.cuda.synchronize()
    print("lane streams on distinct hardware queues:", pool.queue_probe, flush=True)
    alone = replay(plans[:1], streams[:1])
    free = replay(plans, streams, reps=16)

    def barrier(_r):  # every lane waits for every other lane's previous forward
        evs = []
        for s in streams:
            e = torch.cuda.Event()
            e.record(s)
            evs.append(e)
        for s in streams:
            for e in evs:
                s.wait_event(e)
    aligned = replay(plans, streams, reps=16, before_each=barrier)

    def stagger(r):  # lane i starts a quarter forward after lane i - 1, then runs free
        if r == 0:
            for i, s in enumerate(streams):
                with torch.cuda.stream(s):
                    torch.cuda._sleep(int(i * 0.25 * free * 4 * 1e-3 * 2.0e9))
    staggered = replay(plans, streams, reps=16, before_each=stagger)
    # same weights for all four lanes? they are: one PackedUNet.  Control: lane-private copies of the weights would show
    # what "no sharing at all" costs — not built (0.85 GB x 4 is cheap, but the packs are shared by construction)
    print("one lane alone                         %.3f ms per forward" % alone)
    print("four lanes, free-running (bench mode)  %.3f ms per forward" % free)
    print("four lanes, re-aligned every forward   %.3f ms per forward (4 events + 16 waits per forward included)" % aligned)
    print("four lanes, staggered by 1/4 forward   %.3f ms per forward (first-forward sleeps included: %d forwards per lane)" % (staggered, 16))

elif MODE == "controls":
    rows = []
    for lanes, B in ((1, 8), (1, 16), (1, 32), (2, 16), (4, 8), (4, 16), (2, 32)):
        pool = LanePool(lanes) if lanes > 1 else None
        streams = list(pool.streams) if pool else [torch.cuda.Stream()]  # (the legacy stream cannot be captured)
        for conc in ((1, 4) if lanes == 1 else (lanes,)):  # one chain: both tables (the shared-chip one has the big tiles)
            t0 = time.time()
            plans = [lane_plan(i, s, conc, B) for i, s in enumerate(streams)]
            torch.cuda.synchronize()
            hits = [pl.apply_tuning() for pl in plans[:1]]
            ms = replay(plans, streams)
            rows.append((lanes, B, "shared-chip" if conc > 1 else "latency", ms, ms * 8 / B, hits[0]))
            print("lanes %d B %2d table %-11s: %.3f ms per forward = %.3f ms per 8 samples (tuned hits / tuned now / cost-model: %s; %.0f s)" % (
                lanes, B, rows[-1][2], ms, ms * 8 / B, hits[0], time.time() - t0), flush=True)
            for pl in plans:
                pl.close()
            unet._plans.clear()
        if pool:
            pool.close()
    print("\n%5s %3s %-12s %14s %16s %12s" % ("lanes", "B", "table", "ms per forward", "ms per 8 samples", "images/s UNet"))
    for lanes, B, tab, ms, ms8, _ in rows:
        print("%5d %3d %-12s %14.3f %16.3f %12.1f" % (lanes, B, tab, ms, ms8, 8 / (ms8 * 50) * 1e3))

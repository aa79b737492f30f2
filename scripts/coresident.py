"""What does a second resident workgroup per CU buy?  One conv shape, R back-to-back launches per stream (captured), on 1 and
on 4 streams (distinct hardware queues, own buffers / weights / workspace per stream), for tile configurations whose LDS ring
and registers allow one or two workgroups per CU.  Per-launch time alone vs CHIP time per launch with four chains in flight."""
import ctypes as C, math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd
from upgpt_amd import _lib as L
from upgpt_amd.lanes import LanePool
SHAPES = {"c3_M8192": (3, 224, 224, 32, 32), "c3_M2048": (3, 448, 448, 16, 16), "c3_M512": (3, 896, 896, 8, 8),
          "k1_M2048": (1, 448, 448, 16, 16), "k1_M512": (1, 896, 896, 8, 8),
          "v128": (3, 128, 128, 256, 256), "v256": (3, 256, 256, 128, 128), "v512": (3, 512, 512, 64, 64), "v512s": (3, 512, 512, 32, 32)}
CFGS = sys.argv[2:] or ["1x7x4x1k4w3:1", "1x7x4x1k2w3:1", "4x2x2x2k2w3:1", "2x4x4x1k2w3:1", "2x4x2x2k2w3:1", "2x7x2x2k2w3:1", "2x2x2x2k2w3:1", "2x2x2x2k4w3:1"]
name = sys.argv[1] if len(sys.argv) > 1 else "c3_M8192"
ks, cin, cout, H, W = SHAPES[name]
B, NL = 8, 4
R = int(os.environ.get("LAUNCHES", "0")) or (20 if H * W <= 1024 else 3)
pool = LanePool(NL)
print("streams on distinct queues:", pool.queue_probe)
ctxs = [L.get_context(0, lane=i) for i in range(NL)]
lib = ctxs[0].lib
names = [lib.upk_conv_config_name(i).decode() for i in range(lib.upk_conv_num_configs())]
g = torch.Generator().manual_seed(1)
K = ks * ks * cin
sets = []
for i in range(NL):
    x = torch.randn(B, H, W, cin, generator=g).half().cuda()
    w = (torch.randn(cout, cin, ks, ks, generator=g) / math.sqrt(K)).cuda()
    wp, n_pad = ctxs[i].pack_weight(w.contiguous())
    wps = [wp.clone() for _ in range(int(os.environ.get("WEIGHT_COPIES", "8")))]  # (cycled: cold weights, as inside the forward; 1 = warm)
    y = torch.zeros(B, H, W, cout, device="cuda", dtype=torch.float16)
    bias = torch.zeros(n_pad, device="cuda")
    sets.append((x, wps, y, bias, n_pad))
gf = 2.0 * B * H * W * cout * K / 1e9
print("%s: k%d %d -> %d @%dx%d M=%d, %.2f GF per launch" % (name, ks, cin, cout, H, W, B * H * W, gf))


SHARE = os.environ.get("SHARE_WEIGHTS", "0") == "1"  # every lane reads lane 0's weight copies (as the lanes of one model do)


def graph(i, cfg, sk, stream):
    x, wps, y, bias, n_pad = sets[i]
    if SHARE:
        wps = sets[0][1]
    ctx = ctxs[i]
    ds = []
    for r in range(R):
        d = L.ConvDesc()
        d.x1, d.c1, d.ld1 = x.data_ptr(), cin, cin
        d.batch, d.in_h, d.in_w, d.ksize, d.stride = B, H, W, ks, 1
        d.w_packed, d.n_out, d.n_pad, d.bias = wps[r % len(wps)].data_ptr(), cout, n_pad, bias.data_ptr()
        d.y, d.ldy, d.flags = y.data_ptr(), cout, 0
        d.tune_cfg, d.tune_splitk = cfg + 1, sk
        ds.append(d)
    with torch.cuda.stream(stream):
        ctx._chk(lib.upk_graph_begin(ctx.h, stream.cuda_stream))
        for d in ds:
            ctx._chk(lib.upk_conv2d_nhwc_f16(ctx.h, C.byref(d), stream.cuda_stream))
        gh = C.c_void_p(); ctx._chk(lib.upk_graph_end(ctx.h, stream.cuda_stream, C.byref(gh)))
    return gh, ds


def timeit(gs, streams, reps=5):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for (gh, _), s, ctx in zip(gs, streams, ctxs):
                ctx._chk(lib.upk_graph_launch(ctx.h, gh, s.cuda_stream))
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (reps * R * len(gs)) * 1e6)
    return best


print("%-18s %10s %14s %10s %10s" % ("configuration", "alone us", "4 in flight us", "ratio", "TFLOP/s"))
for spec in CFGS:
    cn, sk = spec.split(":")
    if cn not in names:
        print(cn, "unknown"); continue
    cfg = names.index(cn)
    try:
        gs = [graph(i, cfg, int(sk), pool.streams[i]) for i in range(NL)]
    except Exception as e:
        print("%-18s refused (%s)" % (spec, str(e)[:60])); continue
    for (gh, _), s, ctx in zip(gs, pool.streams, ctxs):
        ctx._chk(lib.upk_graph_launch(ctx.h, gh, s.cuda_stream))
    t1 = timeit(gs[:1], pool.streams[:1])
    t4 = timeit(gs, pool.streams)
    print("%-18s %10.2f %14.2f %10.2f %10.0f" % (spec, t1, t4, t4 / t1, gf / t4 * 1e3), flush=True)
    for (gh, _), ctx in zip(gs, ctxs):
        ctx.graph_destroy(gh)

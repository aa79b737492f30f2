"""In-kernel timeline of ONE igemm launch inside the replayed UNet forward graph (dev tool).
   UPK_TL_TARGET=<n-th conv launch of the process> python scripts/timeline_fwd.py
   (needs the stamp build: UPK_CXXFLAGS=-DUPK_TIMELINE, rm upgpt_amd/libupk.so)"""
import contextlib, io, os, sys
import ctypes as C
os.environ["UPK_CXXFLAGS"] = "-DUPK_TIMELINE"
os.environ["UPK_ABLATE"] = "0x200000"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upgpt_amd
from upgpt_amd import synth
with contextlib.redirect_stdout(io.StringIO()):
    model = upgpt_amd.build_model("bbox")
synth.fill_module_(model); model = model.cuda()
pl = model.model.diffusion_model.plan(8, 32, 32, 87, 50, "sampler")
pl.prep.run(); torch.cuda.synchronize()
ctx = pl.ctx
names = {0: "c.entry", 1: "c.stage0 ready", 2: "c.stage1 start", 3: "c.loop end", 4: "c.stores done", 5: "h.reduced", 6: "h.stored",
         8: "l.entry", 9: "l.setup done", 10: "l.prologue issued", 11: "l.stage0 landed", 12: "l.loop end"}
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    sp = s.cuda_stream
    ctx._chk(ctx.lib.upk_graph_begin(ctx.h, sp))
    pl.body.run(sp)
    g = C.c_void_p(); ctx._chk(ctx.lib.upk_graph_end(ctx.h, sp, C.byref(g)))
    for rep in range(3):
        ctx.workspace[-4096:].zero_()
        ctx._chk(ctx.lib.upk_graph_launch(ctx.h, g, sp)); s.synchronize()
        st = ctx.workspace[-4096:].view(torch.int64).cpu().numpy()
        for blk, off in (("first", 0), ("last", 32)):
            t = {k: int(st[off + k]) for k in names if st[off + k]}
            if not t: continue
            t0 = min(t.values())
            print("replay %d block %-5s: " % (rep, blk) + "  ".join("%s +%d" % (names[k], t[k] - t0) for k in sorted(t, key=lambda k: t[k])))

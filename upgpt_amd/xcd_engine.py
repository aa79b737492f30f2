"""Host side of the per-XCD persistent engine (csrc/xcd.hip, include/upk.h upk_xcd_run_f16): the CU grid of a GEMM
phase and the lowering of one SpatialTransformer to a phase list (XcdMixin.xcd_block, mixed into UNetPlan)."""
import ctypes as C
import os

import torch

from . import _lib as L
from . import knobs as K
from ._check import require
from .emitter import Act
from .packing import _rup, head_pad


def xcd_gemm_grid(n, ntiles, K, pair=False, ln=False, lds_bytes=152 * 1024 - 64):
    """(pm, pn, mb, tn, wk) of a per-XCD engine GEMM: the XCD's 32 CUs as a pm x pn grid of (mb rows) x (ntiles / pn column
    tiles); inside a CU the 8 waves as (8 / wk) tile groups of tn tiles x wk slices of K.  Cost model in CU cycles, from
    the engine's in-kernel stamps (scripts/xcd_timeline.py): staging of the CU's rows (one L2 round trip + the bytes at
    ~64 B / clk) + max(weight stream into the CU at ~45 B / clk, MFMA issue at 4 SIMDs x one 16x16x32 per 17 clk, the serial
    chain of the busiest wave: >= 100 clk per chunk with 8 fragments in flight at an L2 latency of ~800 clk) + the LDS
    reduction when K is split.  pair: tiles come in (value, gate) / q|k|v pairs."""
    KC = (K + 31) // 32
    best = None
    for pm in (1, 2, 4, 8, 16, 32):
        mb = _rup((n + pm - 1) // pm, 16)
        if mb > 64:
            continue
        pm_eff = (n + mb - 1) // mb
        tm = mb // 16
        for pn0 in range(1, 32 // pm_eff + 1):
            pn = min(pn0, ntiles // 2 if pair else ntiles)
            tpc = (ntiles + pn - 1) // pn
            if pair and tpc % 2:
                tpc += 1
            pn = (ntiles + tpc - 1) // tpc
            for tn in ((2,) if pair else (1, 2)):
                units = (tpc + tn - 1) // tn
                for wk in (1, 2, 4, 8):
                    wn = 8 // wk
                    if wk > 1 and units > wn:
                        continue
                    kc_per = _rup((KC + wk - 1) // wk, 4)
                    Kpad = kc_per * wk * 32
                    if mb * (Kpad * 2 + 96) > lds_bytes:
                        continue
                    rounds = (units + wn - 1) // wn
                    active = min(8, units * wk)
                    # (measured: the A tile arrives from the shared L2 at ~25 B / clk per CU with all 32 CUs pulling,
                    # a ring fill of 16 KiB of L2-resident weights takes ~1.5 k clk, LayerNorm staging ~1 clk per 50 elements)
                    stage = 1500 + mb * Kpad * 2 / 25.0 + ((1500 + mb * Kpad / 50.0) if ln else 0.0)
                    fill = tpc * 16 * K * 2 / 40.0
                    mfma = tm * tpc * KC * 17 / 4.0
                    per_chunk = max(1500.0 * tn / 16.0, tm * tn * 17.0 * (2 if active > 4 else 1))
                    chain = rounds * kc_per * per_chunk + 1500
                    cost = stage + max(fill, mfma, chain) + (800 if wk > 1 else 0)
                    if best is None or cost < best[0]:
                        best = (cost, pm_eff, pn, mb, tn, wk)
    return None if best is None else best[1:]


class XcdMixin:
    """UNetPlan.xcd_block: see the module docstring."""

    def xcd_block(self, P, Lr, x):
        """The whole SpatialTransformer (attention.py:250-261) as ONE launch of the per-XCD engine (include/upk.h
        upk_xcd_run_f16; csrc/xcd.hip): GroupNorm -> proj_in -> [LN1 -> q|k|v -> self-attention -> to_out + t0] ->
        [LN2 -> to_q -> attention over the precomputed context K / V -> to_out + t1] -> [LN3 -> GEGLU] ->
        ff.net.2 o proj_out + x: ten phases, XCD-local barriers between them.  Returns the output Act, or None when the
        engine is off / does not take the shape (the caller then emits the launch chain)."""
        n = Lr.name
        wx = self.pk.w.get(n + ".xcd")
        if K.XCD == "0" or wx is None or n not in self.kv:
            return None
        B, HW, M, C_ = x.B, x.H * x.W, x.M, x.C
        if K.XCD != "1" and (B % 8 or HW > K.XCD_MAXN):
            return None
        if x.ld != C_ or HW % 4 or self.ctx.num_cus != 256:
            return None
        heads, dh = Lr.heads, Lr.dhead
        dp = head_pad(dh)
        hd, inner = heads * dp, wx["inner"]
        kc, vtc, cld = self.kv[n]
        vt_ld = _rup(HW, 32)
        A = lambda cols, zero=False: self.alloc(M, cols, zero=zero)
        xn, t0, qk, a1, t1, q2, a2, t2, hg, y = (A(C_), A(C_), A(2 * hd), A(hd), A(C_), A(hd), A(hd), A(C_), A(inner),
                                                  A(C_))
        vt = self.alloc(B, heads, dp, vt_ld, zero=True)
        cs = float(dh ** -0.5 * 1.4426950408889634)
        ph = []

        def gemm(a, k1, pw, y_, ldy, n_out, *, ln=0, a2=None, k2=0, lda2=0, res=None, epi=L.XE_PLAIN, lda=None):
            q = L.XPhase()
            q.kind, q.n = L.XP_GEMM, HW
            q.a, q.lda, q.k1 = a.data_ptr(), (lda or a.shape[-1]), k1
            if a2 is not None:
                q.a2, q.lda2, q.k2 = a2.data_ptr(), lda2, k2
            require(k1 + k2 == pw.k, lambda: repr(("xcd K mismatch", k1, k2, pw.k)), ValueError)
            q.w, q.ntiles, q.n_out = pw.w.data_ptr(), pw.ntiles, n_out
            if pw.bias is not None:
                q.bias = pw.bias.data_ptr()
            if res is not None:
                q.res, q.ldres = res.data_ptr(), res.shape[-1]
            q.y, q.ldy, q.epi, q.ln, q.eps = y_.data_ptr(), ldy, epi, ln, 1e-5
            if ln:
                q.colsum = pw.colsum.data_ptr()
            pair = epi != L.XE_PLAIN
            grid = xcd_gemm_grid(HW, pw.ntiles, pw.k, pair=pair, ln=bool(ln))
            if grid is None:
                return None
            q.pm, q.pn, q.mb, q.tn, q.wk = grid
            ov = os.environ.get("UPGPT_XCD_GRID")  # "pm,pn,mb,tn,wk" forced on every GEMM phase (experiments)
            if ov:
                q.pm, q.pn, q.mb, q.tn, q.wk = (int(v) for v in ov.split(","))
            return q

        g = L.XPhase()
        g.kind, g.n, g.a, g.lda, g.k1 = L.XP_GN, HW, x.t.data_ptr(), x.ld, C_
        g.groups, g.eps, g.silu, g.y, g.ldy = 32, 1e-6, 0, xn.data_ptr(), C_
        ph.append(g)
        ph.append(gemm(xn, C_, wx["proj_in"], t0, C_, C_))
        q = gemm(t0, C_, wx["qkv"], qk, 2 * hd, 2 * hd, ln=1, epi=L.XE_QKV)
        if q is not None:
            q.vt, q.vt_ld, q.heads, q.dp, q.vtile0 = vt.data_ptr(), vt_ld, heads, dp, 2 * hd // 16
        ph.append(q)
        at = L.XPhase()
        at.kind, at.n, at.a, at.lda = L.XP_ATTN, HW, qk.data_ptr(), 2 * hd
        at.kk, at.ldk, at.koff, at.kbs, at.nkv = qk.data_ptr(), 2 * hd, hd, HW * 2 * hd, HW
        at.vv, at.vt_ld, at.vbs = vt.data_ptr(), vt_ld, heads * dp * vt_ld
        at.y, at.ldy, at.heads, at.dp, at.scale_log2 = a1.data_ptr(), hd, heads, dp, cs
        ph.append(at)
        ph.append(gemm(a1, hd, wx["out1"], t1, C_, C_, res=t0))
        ph.append(gemm(t1, C_, wx["q2"], q2, hd, hd, ln=1))
        ax = L.XPhase()
        ax.kind, ax.n, ax.a, ax.lda = L.XP_ATTN, HW, q2.data_ptr(), hd
        ax.kk, ax.ldk, ax.koff, ax.kbs, ax.nkv = kc.t.data_ptr(), kc.ld, 0, self.n_ctx * kc.ld, self.n_ctx
        ax.vv, ax.vt_ld, ax.vbs = vtc.data_ptr(), cld, heads * dp * cld
        ax.y, ax.ldy, ax.heads, ax.dp, ax.scale_log2 = a2.data_ptr(), hd, heads, dp, cs
        ph.append(ax)
        ph.append(gemm(a2, hd, wx["out2"], t2, C_, C_, res=t1))
        ph.append(gemm(t2, C_, wx["geglu"], hg, inner, inner, ln=1, epi=L.XE_GEGLU))
        ph.append(gemm(hg, inner, wx["ffout"], y, C_, C_, a2=t2, k2=C_, lda2=C_, res=x.t, lda=inner))
        verbose = os.environ.get("UPGPT_XCD_VERBOSE", "0") == "1"
        if any(q is None for q in ph):
            if verbose:
                print("[xcd] %s: no CU grid for phase %d" % (n, [q is None for q in ph].index(True)))
            return None
        for i, q in enumerate(ph):
            if self.lib.upk_xcd_phase_check(self.hctx, C.byref(q)) != 0:
                if verbose:
                    print("[xcd] %s: phase %d refused: %s" % (n, i, (self.lib.upk_last_error(self.hctx) or b"").decode()))
                return None
        for i, q in enumerate(ph):  # the next GEMM phase: its weights are prefetched while phase i runs
            q.nx = next((k for k in range(i + 1, len(ph)) if ph[k].kind == L.XP_GEMM), -1)
        arr = (L.XPhase * len(ph))(*ph)
        dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev)
        self.bufs.append(dev)
        if getattr(self, "xcd_sync", None) is None:
            self.xcd_sync = self.alloc(self.lib.upk_xcd_sync_bytes(), dtype=torch.uint8, zero=True)
        sync = self.xcd_sync
        fn, h, chk = self.lib.upk_xcd_run_f16, self.hctx, self._chk
        base, nph, sz = dev.data_ptr(), len(ph), C.sizeof(L.XPhase)
        if K.XCD_SPLIT:
            def run(s):
                for i in range(nph):
                    chk(fn(h, base + i * sz, 1, B, sync.data_ptr(), s))
        else:
            def run(s):
                chk(fn(h, base, nph, B, sync.data_ptr(), s))
        if os.environ.get("UPGPT_XCD_KEEP", "0") == "1":  # (scripts/xcd_debug.py compares every intermediate)
            self.__dict__.setdefault("xcd_dbg", {})[n] = dict(x=x.t, xn=xn, t0=t0, qk=qk, vt=vt, a1=a1, t1=t1, q2=q2, a2=a2,
                                                              t2=t2, hg=hg, y=y, kc=kc.t, vtc=vtc)
        P.add(run, x, wx, kc, vtc, dev, sync, arr, cls="igemm_k1", label="xcd M%d C%d d%d" % (M, C_, dp))
        fl = 2 * M * sum(wx[k].n * wx[k].k for k in ("proj_in", "out1", "q2", "out2", "geglu", "ffout"))
        fl += 2 * M * C_ * 3 * heads * dh  # (q | k | v at the real head width)
        P.igemm_flops += fl
        P.flops[-1] = fl
        P.attn_flops += 4 * B * heads * HW * (HW + self.n_ctx) * dh
        return Act(y, B, x.H, x.W, C_)


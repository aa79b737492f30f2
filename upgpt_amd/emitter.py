"""Shared emission helpers: activations, flat programs of launches, and the Emitter that turns layer records into
upk_* launches (conv / GEMM with its fold decisions, GroupNorm / LayerNorm, the fused row-chain kernels, attention)."""
import ctypes as C
import os

import torch

from . import _lib as L
from . import knobs as K
from ._check import require
from .packing import PW, Packer, _rup, head_pad
from .tuning import TUNE_CACHE, TUNE_CACHE_LANES


class Act:
    """[B*H*W, ld] fp16 activation (C valid channels)."""
    __slots__ = ("t", "B", "H", "W", "C", "gn_src", "ln_src")

    def __init__(self, t, B, H, W, C):
        self.t, self.B, self.H, self.W, self.C = t, B, H, W, C
        self.ln_src = None  # ConvDesc of the launch that wrote this tensor (it may leave LayerNorm row sums)
        self.gn_src = None  # (producer ConvDesc, stats buffer) when the producer may have left GroupNorm partials

    @property
    def ld(self):
        return self.t.shape[-1]

    @property
    def M(self):
        return self.B * self.H * self.W


class Program:
    """A flat list of launches; each op is a callable taking the stream pointer."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.ops = []
        self.cls = []
        self.labels = []
        self.keep = []
        self.flops = []  # algorithmic FLOPs of each op (conv / GEMM launches; 0 elsewhere)
        self.meta = []   # ConvDesc of a upk_conv2d launch (its tuned configuration names the kernel instantiation), else None
        self.igemm_flops = 0
        self.attn_flops = 0
        self.n_launch = 0

    def run(self, stream=None, skip=(), skip_idx=()):
        """skip: op classes / skip_idx: op indices to leave out (ablation timing only: results are garbage)."""
        s = self.ctx._s() if stream is None else stream
        if skip or skip_idx:
            for i, (op, cls) in enumerate(zip(self.ops, self.cls)):
                if cls not in skip and i not in skip_idx:
                    op(s)
            return
        for op in self.ops:
            op(s)

    def add(self, fn, *keep, cls="other", label=None):
        self.ops.append(fn)
        self.flops.append(0)
        self.meta.append(None)
        self.cls.append(cls)
        self.labels.append(label or cls)
        self.keep.extend(keep)
        self.n_launch += 1


class Emitter:
    """Shared emission helpers (conv / gemm / norms / attention) for both engines."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.lib = ctx.lib
        self.hctx = ctx.h
        self.dev = ctx.device
        self.bufs = []
        self.convs = []  # (ConvDesc, shape-signature) of every emitted conv, for autotuning
        TUNE_CACHE.bind(self.lib)  # (the emitters consult the cache while they lower the network)

    def apply_tuning(self, cache=None, tune_missing=False, reps=None):
        """Pins each conv launch to the (tile config, split-K) stored in the tuning cache;
        with tune_missing=True unknown shapes are timed on the device first
        (upk_conv_autotune) and added to the cache.  Returns (#hits, #tuned, #missing)."""
        overlay = None
        if cache is None:
            cache = TUNE_CACHE
            if L.concurrency() > 1 and os.environ.get("UPGPT_LANES_TUNING", "1") == "1":
                overlay = TUNE_CACHE_LANES  # (several batches in flight: the throughput-tuned choices come first)
                overlay.bind(self.lib)
        cache.bind(self.lib)

        def lookup(c, key):
            ent = c.get(key)
            if ent is None and not tune_missing and key.endswith("_gs"):
                ent = c.get(key[:-3])  # (statistics by-product armed on a shape that was tuned without it)
            if ent is None and not tune_missing and not key.endswith("_gs"):
                ent = c.get(key + "_gs")  # (tuned with the statistics by-product armed; the choice is valid without)
            if ent is None and not tune_missing and key.endswith("_lnr"):
                ent = c.get(key[:-4])  # (tuned as a plain GEMM; usable only if that choice does not split K)
                if ent is not None and int(ent[1]) != 1 and not self._is_as(int(ent[0])):
                    ent = None
            return ent

        hits = tuned = missing = 0
        for d, key in self.convs:
            ent = lookup(overlay, key) if overlay is not None else None
            if ent is None:
                ent = lookup(cache, key)
            if ent is None and tune_missing:
                cfg, sk, best_us, dflt_us = self.ctx.conv_autotune(d, reps or int(os.environ.get("UPGPT_TUNE_REPS", "5")))
                ent = cache.put(key, cfg, sk, best_us, dflt_us)
                tuned += 1
            elif ent is not None:
                hits += 1
            else:
                missing += 1
            if ent is not None:
                d.tune_cfg, d.tune_splitk = int(ent[0]) + 1, int(ent[1])
        if tuned and cache.dirty and os.environ.get("UPGPT_TUNE_SAVE"):
            # shapes outside the shipped table (another batch / latent size) were just measured on this device
            # (UPGPT_AUTOTUNE=1): UPGPT_TUNE_SAVE=<path> keeps them for the next process (load with UPGPT_TUNE_FILE=<path>)
            cache.save(os.environ["UPGPT_TUNE_SAVE"])
        return hits, tuned, missing

    def link_weight_prefetch(self, prog, wrap=True):
        """Every conv / Linear launch of `prog` is told the packed weight of the NEXT one (include/upk.h pf_next): its
        idle MFMA waves pull those lines into the memory-side cache while their own first ring stage is in flight, so the next
        launch does not start on cold weights.  UPGPT_WEIGHT_PREFETCH: "auto" = only while ONE batch has the
        chip to itself — measured: forward 2.83 -> 2.76 ms alone, 1.461 -> 1.471 ms per forward with four in flight (the
        shared chip has no idle fabric slots to hide the touches in) — "0" never, "1" always.  wrap: the last launch prefetches
        for the first one — the program is replayed step after step.  Returns the number of links."""
        mode = K.WEIGHT_PREFETCH
        if mode == "0" or (mode == "auto" and L.concurrency() > 1):
            return 0
        ds = [d for d in prog.meta if d is not None and getattr(d, "_w_bytes", 0)]
        n = 0
        ahead = max(1, K.WEIGHT_PREFETCH_AHEAD)
        for i, d in enumerate(ds):
            j = i + ahead
            nxt = ds[j] if j < len(ds) else (ds[j % len(ds)] if wrap and len(ds) > ahead else None)
            if nxt is None or nxt.w_packed == d.w_packed:
                continue
            d.pf_next, d.pf_bytes = nxt.w_packed, min(nxt._w_bytes, K.WEIGHT_PREFETCH_MAX)
            n += 1
        return n

    def _is_as(self, cfg):
        """Whether configuration `cfg` belongs to the A-stationary family (their second tuning slot is output-column
        passes per workgroup, not a split-K factor)."""
        return 0 <= cfg < self.lib.upk_conv_num_configs() and self.lib.upk_conv_config_name(cfg).decode().startswith("as")

    def alloc(self, *shape, dtype=torch.float16, zero=False):
        t = (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=self.dev)
        self.bufs.append(t)
        return t

    def _chk(self, rc):
        if rc != 0:
            self.ctx._chk(rc)

    @staticmethod
    def conv_key(M, n_pad, c1, c2, ks, stride, flags, has_res, has_rowvec, has_vt, ln, gs=False, ka=0):
        """Shape signature of one conv/GEMM launch = key of the tuning cache."""
        return "M%d_N%d_C%d+%d_k%ds%d_f%x_r%d%d%d%s%s%s" % (M, n_pad, c1, c2, ks, stride, flags, has_res, has_rowvec,
                                                           has_vt, "_ln" if ln else "", "_gs" if gs else "",
                                                           "_ka%d" % ka if ka else "")

    def fold_skip(self, hN, pw_main, pw_skip, x, skip):
        """Whether a ResBlock's 1x1 skip projection rides along its second conv as an appended K segment
        (include/upk.h x3/x4).  UPGPT_SKIP_FOLD=0/1 forces it; by default the tuning cache decides: fused launch vs
        conv (with residual) + skip conv, both measured by scripts/tune.py; unknown shapes keep two launches."""
        mode = os.environ.get("UPGPT_SKIP_FOLD", "auto")
        if mode != "auto":
            return mode == "1"
        c3, c4 = _rup(x.C, 32), (_rup(skip.C, 32) if skip is not None else 0)
        base = (hN.M, pw_main.n_pad, _rup(hN.C, 32), 0, pw_main.ksize, 1, 0)
        def tuned(key):  # (a launch whose output feeds a GroupNorm is tuned under its "_gs" name)
            return TUNE_CACHE.get(key + "_gs") or TUNE_CACHE.get(key)

        e_f = tuned(self.conv_key(*base, False, False, False, False, ka=c3 + c4))
        e_m = tuned(self.conv_key(*base, True, False, False, False))
        e_s = tuned(self.conv_key(hN.M, pw_skip.n_pad, c3, c4, 1, 1, 0, False, False, False, False))
        if e_f is None or e_m is None or e_s is None:
            return False
        return e_f[2] < e_m[2] + e_s[2]

    def fold_ff_out(self, ff, t2, pw_ff):
        """Whether a SpatialTransformer's last two linear maps — ff.net.2 (+ residual t2, attention.py:215) and
        proj_out (+ residual x_in, attention.py:259-261), nothing but a reshape between them — run as one GEMM over
        [ff | t2] with the pre-multiplied weight [P F2 | P].  UPGPT_FFOUT_FOLD=0/1 forces it; by default the tuning
        cache decides (fused launch vs the two launches); unknown shapes keep two launches."""
        mode = os.environ.get("UPGPT_FFOUT_FOLD", "auto")
        if mode != "auto":
            return mode == "1"

        def tuned(key):
            return TUNE_CACHE.get(key + "_gs") or TUNE_CACHE.get(key)

        M, n_pad, c_ff, c_t = ff.M, pw_ff.n_pad, _rup(ff.C, 32), _rup(t2.C, 32)
        e_f = tuned(self.conv_key(M, n_pad, c_ff, 0, 1, 1, 0, True, False, False, False, ka=c_t))
        e_a = tuned(self.conv_key(M, n_pad, c_ff, 0, 1, 1, 0, True, False, False, False))
        e_b = tuned(self.conv_key(M, n_pad, c_t, 0, 1, 1, 0, True, False, False, False))
        if e_f is None or e_a is None or e_b is None:
            return False
        return e_f[2] < e_a[2] + e_b[2]

    def ln_linear(self, P, x, name, norm, flags=0, **kw):
        """LayerNorm `norm` followed by the Linear `name`: either one launch with the norm folded into the
        GEMM (weights packed as name + "_ln") or LayerNorm launch + plain GEMM, whichever the tuning
        cache says is faster for this shape (the fold costs VALU work in the GEMM's MFMA waves and rules
        out the classic / K-split tile configurations; a LayerNorm launch costs ~3.8 us under replay).
        UPGPT_LN_FOLD=0/1 forces the choice (scripts/tune.py measures both).

        When the launch that produced `x` can leave the row sums (include/upk.h ln_rows_out: plain epilogue, no
        split-K), the fold takes them from there instead: no LayerNorm launch, no statistics work in the GEMM, any
        tile configuration (UPGPT_LN_ROWS=0 switches this off)."""
        prod = getattr(x, "ln_src", None)
        if K.LN_ROWS and prod is not None and not prod.ln_rows_out and x.C == x.ld:
            rows = self.alloc(8, x.M, 2, dtype=torch.float32)
            prod.ln_rows_out = rows.data_ptr()
            out = kw.pop("out", None)
            pw = self.pk.w[name + "_ln"]
            if out is None:  # (both programs write the same buffer)
                require(pw.n_out % 32 == 0, "folded-LayerNorm output width must be a multiple of 32", ValueError)
                out = Act(self.alloc(x.M, pw.n_out), x.B, x.H, x.W, pw.n_out)
            alt = Program(self.ctx)
            self._ln_linear_plain(alt, x, name, norm, flags, out=out, **kw)
            return self.conv(P, x, pw, flags=flags, ln_eps=1e-5, lnr=rows, lnr_alt=alt, out=out, **kw)
        return self._ln_linear_plain(P, x, name, norm, flags, **kw)

    def _ln_linear_plain(self, P, x, name, norm, flags=0, **kw):
        w, v = self.pk.w, self.pk.v
        mode = os.environ.get("UPGPT_LN_FOLD", "auto")
        fold = mode != "0"
        if mode == "auto":
            pw = w[name]
            args = (x.M, pw.n_pad, _rup(x.C, 32), 0, 1, 1, flags, False, False, "vt" in kw and kw["vt"] is not None)
            e_ln = TUNE_CACHE.get(self.conv_key(*args, True))
            e_pl = TUNE_CACHE.get(self.conv_key(*args, False))
            if e_ln is not None and e_pl is not None:
                fold = e_ln[2] < e_pl[2] + K.LN_LAUNCH_US
        if fold:
            return self.conv(P, x, w[name + "_ln"], flags=flags, ln_eps=1e-5, **kw)
        return self.conv(P, self.layernorm(P, x, *v[norm]), w[name], flags=flags, **kw)

    def conv(self, P, x1, pw, *, x2=None, stride=1, flags=0, residual=None, rowvec=None, rv_bs=0, rv_ss=0,
             step=None, out=None, vt=None, nchw_out=None, out_f32=None, spatial=None, ln_eps=None,
             gn_stats=False, append=None, gn=None, lnr=None, lnr_alt=None):
        """Emits one upk_conv2d_nhwc_f16. Returns the output Act (fp16) unless nchw_out /
        out_f32 is given.

        gn = (gamma, beta, eps, silu, ws[, sole]): x1 | x2 are UN-normalised; the GroupNorm(+SiLU) in front of this conv
        (openaimodel.py:255-275, attention.py:250-256) is emitted first (Emitter.groupnorm: apply-only when the producer
        left the statistics, inside the producer's split-K reduce pass when it has one)."""
        B, H, W = spatial if spatial is not None else (x1.B, x1.H, x1.W)
        ks = pw.ksize
        kw_all = dict(stride=stride, flags=flags, residual=residual, rowvec=rowvec, rv_bs=rv_bs, rv_ss=rv_ss, step=step,
                      out=out, vt=vt, nchw_out=nchw_out, out_f32=out_f32, spatial=spatial, ln_eps=ln_eps,
                      gn_stats=gn_stats, append=append, lnr=lnr, lnr_alt=lnr_alt)
        if gn is not None:
            return self.conv(P, self.groupnorm(P, x1, *gn[:5], x2=x2, sole=len(gn) > 5 and gn[5]), pw, **kw_all)
        ups = bool(flags & L.F_UPSAMPLE2X)
        HL, WL = (2 * H, 2 * W) if ups else (H, W)
        if flags & L.F_PAD_ASYM:
            Ho, Wo = (HL + 1 - 3) // 2 + 1, (WL + 1 - 3) // 2 + 1
        else:
            pad = 1 if ks == 3 else 0
            Ho, Wo = (HL + 2 * pad - ks) // stride + 1, (WL + 2 * pad - ks) // stride + 1
        M = B * Ho * Wo
        d = L.ConvDesc()
        d.x1 = x1.t.data_ptr()
        d.c1 = _rup(x1.C, 32)
        d.ld1 = x1.ld
        if x2 is not None:
            d.x2 = x2.t.data_ptr()
            d.c2 = _rup(x2.C, 32)
            d.ld2 = x2.ld
        require(d.c1 + d.c2 == pw.k_packed, lambda: repr(("K mismatch", d.c1, d.c2, pw.k_packed)), ValueError)
        require(d.c1 <= x1.ld and (x2 is None or d.c2 <= x2.ld), "conv: padded channel count exceeds the row stride of its source", ValueError)
        d.batch, d.in_h, d.in_w = B, H, W
        d.ksize, d.stride = ks, stride
        d.w_packed = pw.w.data_ptr()
        d._w_bytes = int(pw.w.numel()) * 2  # (host-side note for link_weight_prefetch; not part of the C struct)
        d.n_out, d.n_pad = pw.n_out, pw.n_pad
        if pw.bias is not None:
            d.bias = pw.bias.data_ptr()
        phased = K.UPS_PHASES and bool(flags & L.F_UPSAMPLE2X) and pw.w_phase is not None and x2 is None
        if phased:
            d.w_phase = pw.w_phase.data_ptr()
        if residual is not None:
            d.residual = residual.t.data_ptr()
            d.ld_res = residual.ld
        if rowvec is not None:
            d.rowvec = rowvec.data_ptr()
            d.rv_batch_stride, d.rv_step_stride = rv_bs, rv_ss
        if step is not None:
            d.step = step.data_ptr()
        ret = None
        if nchw_out is not None:
            d.y = nchw_out.data_ptr()
            d.ldy = 0
            flags |= L.F_OUT_NCHW_F32
        elif out_f32 is not None:
            d.y = out_f32.data_ptr()
            d.ldy = out_f32.shape[-1]
            flags |= L.F_OUT_F32
        else:
            if out is None:
                ld = pw.n_out if pw.n_out % 32 == 0 else _rup(pw.n_out, 32)  # (a consumer conv reads 32-channel chunks)
                out = Act(self.alloc(M, ld, zero=(ld != pw.n_out)), B, Ho, Wo, pw.n_out)
            d.y = out.t.data_ptr()
            d.ldy = out.ld
            ret = out
        if vt is not None:
            d.vt = vt["t"].data_ptr()
            d.vt_from, d.vt_heads, d.vt_dhead = vt["from"], vt["heads"], vt["dhead"]
            d.vt_ld, d.vt_tokens = vt["ld"], vt["tokens"]
        d.flags = flags
        if gn_stats and ret is not None and vt is None and pw.n_out % 8 == 0 and pw.n_out % 32 == 0:
            # if this launch splits K, its reduce pass also writes the GroupNorm partials of the output
            # (include/upk.h gn_stats_ws); the GroupNorm that reads `ret` then runs its apply pass only
            # (armed by the consuming groupnorm(): a by-product nobody reads costs epilogue time and would mislead
            # the tuner's credit for the saved gn_stats launch)
            ret.gn_src = (d, len(self.convs))
        if ret is not None and vt is None and not (flags & (L.F_GEGLU | L.F_SILU)):
            ret.ln_src = d  # (a LayerNorm-folded consumer may ask this launch for the row statistics, see ln_linear)
        if lnr is not None:  # folded LayerNorm with the row statistics from x1's producer (include/upk.h ln_rows_*)
            d.ln_rows_in = lnr.data_ptr()
            d.ln_rows_slots = 1  # (set from the producer's answer when the program runs)
        if ln_eps is not None:  # x1 is the un-normalised residual stream; pw was packed with ln=...
            require(pw.ln_colsum is not None and x2 is None and ks == 1, "folded LayerNorm needs a single-source 1x1 launch with an '_ln' packed weight", ValueError)
            d.ln_colsum = pw.ln_colsum.data_ptr()
            d.ln_eps = float(ln_eps)
            d.ln_dim = x1.C
        x3 = x4 = None
        if append is not None:  # appended 1x1 K segment over (x3 | x4) at the output pixel; pw from Packer.append_1x1
            x3, x4 = append
            require(stride == 1 and not ups and (x3.B, x3.H, x3.W) == (B, Ho, Wo), "appended 1x1 segment: sources must have the output's spatial dims (stride 1, no upsample)", ValueError)
            d.x3, d.c3, d.ld3 = x3.t.data_ptr(), _rup(x3.C, 32), x3.ld
            if x4 is not None:
                d.x4, d.c4, d.ld4 = x4.t.data_ptr(), _rup(x4.C, 32), x4.ld
            require(d.c3 + d.c4 == pw.k_append, lambda: repr(("appended K mismatch", d.c3, d.c4, pw.k_append)), ValueError)
        else:
            require(not pw.k_append, "weight was packed with an appended segment but the launch has none", ValueError)
        key = self.conv_key(M, pw.n_pad, d.c1, d.c2, ks, stride, flags, residual is not None, rowvec is not None,
                            vt is not None, ln_eps is not None and lnr is None, ka=d.c3 + d.c4)
        if lnr is not None:
            # its own entry: the library refuses split-K for any folded LayerNorm, so a (config, split-K > 1) pair tuned
            # for the plain GEMM of the same shape (proj_in vs attn2.q when hd == C ...) must never be pinned on it, and
            # its sk = 1-only autotune result must not pessimise the plain GEMM either (apply_tuning falls back to the
            # plain entry only when that one does not split K)
            key += "_lnr"
        if phased:
            key += "_ph"
        self.convs.append((d, key))
        fn, h, ref = self.lib.upk_conv2d_nhwc_f16, self.hctx, C.byref(d)
        chk = self._chk
        keep = (d, pw, x1, x2, x3, x4, residual, rowvec, out, nchw_out, out_f32, vt)
        if lnr is not None:
            # the producer of x1 leaves the LayerNorm row sums when its (tuned) launch can (plain epilogue, no split-K,
            # M x N-split tile); otherwise the alternative program runs: LayerNorm launch / in-kernel fold
            prod, alt = x1.ln_src, lnr_alt
            ask = self.lib.upk_conv_ln_rows

            def run_lnr(s):
                slots = C.c_int(0)
                chk(ask(h, C.byref(prod), C.byref(slots)))
                if slots.value > 0:
                    d.ln_rows_slots = slots.value
                    chk(fn(h, ref, s))
                else:
                    alt.run(s)

            P.add(run_lnr, *keep, lnr, prod, alt, cls="igemm_k%d" % ks, label=key)
        else:
            P.add(lambda s: chk(fn(h, ref, s)), *keep, cls="igemm_k%d" % ks, label=key)
        P.igemm_flops += 2 * M * pw.n_real * pw.k_real
        P.flops[-1] = 2 * M * pw.n_real * pw.k_real
        P.meta[-1] = d
        return ret

    class GnProvider:
        """A launch other than upk_conv2d_nhwc_f16 that leaves the per-(row block, channel) GroupNorm partials of its
        output (mode 2 of include/upk.h gn_stats_ws): (stats buffer, nblk, ld) are fixed when it is emitted."""
        def __init__(self, sws, nblk, ld):
            self.sws, self.nblk, self.ld = sws, nblk, ld

    def _arm_gn_sources(self, acts):
        """Arms the producer launch of every source Act to leave the GroupNorm partial sums of its output
        (include/upk.h gn_stats_ws).  Returns [(producer ConvDesc, stats buffer)] or None when a source has no such
        producer / a concat source is known to split K (per-group partials cannot be combined across the seam)."""
        srcs = [getattr(a, "gn_src", None) for a in acts]
        if any(sr is None for sr in srcs):
            return None
        if len(acts) > 1:
            if os.environ.get("UPGPT_GN_2SRC", "1") != "1":
                return None
            for sr in srcs:
                if isinstance(sr, Emitter.GnProvider):
                    continue  # (never split K)
                key = self.convs[sr[1]][1]
                e = TUNE_CACHE.get(key) or TUNE_CACHE.get(key[:-3] if key.endswith("_gs") else key + "_gs")
                if e is None or (e[1] != 1 and not self._is_as(int(e[0]))):  # (as*: second slot = passes per workgroup)
                    return None
        armed = []
        for act in acts:
            if isinstance(act.gn_src, Emitter.GnProvider):
                armed.append((act.gn_src, act.gn_src.sws))
                continue
            d = act.gn_src[0]
            if not d.gn_stats_ws:  # arm the producer and rename its tuning key
                ci = act.gn_src[1]
                cap = max(32, (act.H * act.W) // 64)  # (more than 32 row blocks per sample: folded by a finalize launch)
                sws = self.alloc(self.ctx.gn_stats_floats(act.B, d.n_pad, cap), dtype=torch.float32)
                d.gn_stats_ws, d.gn_groups, d.gn_stats_cap = sws.data_ptr(), 32, cap
                require(self.convs[ci][0] is d, "conv list out of sync with GroupNorm producers", RuntimeError)
                self.convs[ci] = (d, self.convs[ci][1] + "_gs")
                act.gn_src = (d, ci, sws)
            armed.append((d, act.gn_src[2]))
        return armed

    def groupnorm(self, P, x1, gamma, beta, eps, silu, ws, x2=None, sole=False):
        """sole: nothing but this GroupNorm reads x1 (its producer may then skip writing it, see gno_skip_y)."""
        Cc = x1.C + (x2.C if x2 is not None else 0)
        y = Act(self.alloc(x1.M, Cc), x1.B, x1.H, x1.W, Cc)
        fn, h, chk = self.lib.upk_groupnorm_nhwc_f16, self.hctx, self._chk
        a = (x1.t.data_ptr(), x1.C, x1.ld, x2.t.data_ptr() if x2 is not None else None, x2.C if x2 is not None else 0,
             x2.ld if x2 is not None else 0, x1.B, x1.H * x1.W, 32, gamma.data_ptr(), beta.data_ptr(), float(eps),
             int(bool(silu)), y.t.data_ptr(), y.ld)
        armed = self._arm_gn_sources([x1] if x2 is None else [x1, x2])
        if armed is None:
            P.add(lambda s: chk(fn(h, *a, ws.data_ptr(), s)), x1, x2, gamma, beta, y, ws, cls="groupnorm",
                  label="gn M%d C%d silu%d 2pass" % (x1.M, Cc, int(bool(silu))))
        else:
            # the producer conv(s) may have left the partial statistics of the input in their own buffers: per-group
            # partials from a split-K reduce pass (single source only) or per-(M tile, channel) partials from an
            # unsplit epilogue (every source of a concat must have them); decided by the tuned / cost-model choice
            # at the time the program runs or is captured
            fused_fn, apply_fn = self.lib.upk_conv_gn_fused, self.lib.upk_groupnorm_apply_nhwc_f16
            fin_fn = self.lib.upk_groupnorm_finalize_f32
            mine = None
            if K.GN_REDUCE_APPLY and x2 is None and not isinstance(armed[0][0], Emitter.GnProvider) and not armed[0][0].gno_y:
                # a producer that splits K normalises in its reduce pass (include/upk.h gno_*): this op then launches nothing
                mine = armed[0][0]
                mine.gno_gamma, mine.gno_beta, mine.gno_eps = gamma.data_ptr(), beta.data_ptr(), float(eps)
                mine.gno_silu, mine.gno_y, mine.gno_ld, mine.gno_skip_y = int(bool(silu)), y.t.data_ptr(), y.ld, int(bool(sole))

            def run(s):
                info = []
                for d, sws in armed:
                    if isinstance(d, Emitter.GnProvider):
                        info.append((2, d.nblk, d.ld, sws.data_ptr()))
                        continue
                    mode, nblk = C.c_int(0), C.c_int(0)
                    chk(fused_fn(h, C.byref(d), C.byref(mode), C.byref(nblk)))
                    info.append((mode.value if mode.value != 3 or d is mine else 0, nblk.value, d.n_pad, sws.data_ptr()))
                if len(info) == 1 and info[0][0] == 3:
                    return
                if len(info) == 1 and info[0][0] == 2 and info[0][1] > 32:
                    # long feature maps (VAE decoder): the producer's channel partials are folded per (sample, group)
                    # first — one small launch instead of a statistics pass over the tensor
                    m, nb, ld, p1 = info[0]
                    chk(fin_fn(h, p1, nb, ld, x1.B, x1.H * x1.W, x1.C, 32, ws.data_ptr(), s))
                    chk(apply_fn(h, *a, ws.data_ptr(), 1, 0, 0, None, 0, 0, s))
                    return
                if len(info) == 1 and info[0][0]:
                    m, nb, ld, p1 = info[0]
                    chk(apply_fn(h, *a, p1, m, nb, ld, None, 0, 0, s))
                elif len(info) == 2 and info[0][0] == 2 and info[1][0] == 2 and max(info[0][1], info[1][1]) <= 32:
                    chk(apply_fn(h, *a, info[0][3], 2, info[0][1], info[0][2], info[1][3], info[1][1], info[1][2], s))
                else:
                    chk(fn(h, *a, ws.data_ptr(), s))

            P.add(run, x1, x2, gamma, beta, y, ws, armed, cls="groupnorm",
                  label="gn M%d C%d silu%d" % (x1.M, Cc, int(bool(silu))))
        P.n_launch += 1  # stats + apply
        return y

    def layernorm(self, P, x, gamma, beta, eps=1e-5):
        y = Act(self.alloc(x.M, x.C), x.B, x.H, x.W, x.C)
        fn, h, chk = self.lib.upk_layernorm_f16, self.hctx, self._chk
        a = (x.t.data_ptr(), x.ld, x.M, x.C, gamma.data_ptr(), beta.data_ptr(), float(eps), y.t.data_ptr(), y.ld)
        P.add(lambda s: chk(fn(h, *a, s)), x, gamma, beta, y, cls="layernorm", label="ln M%d C%d" % (x.M, x.C))
        return y

    def mlp_rows(self, M):
        """Rows per workgroup of the fused feed-forward kernel: 64 while that still gives every CU a workgroup."""
        if K.MLP_ROWS in (32, 64):
            return K.MLP_ROWS
        return 64 if M // 64 >= self.ctx.num_cus else 32

    def geglu_mlp(self, P, t2, x_in, pw1, pw2, gn_stats=True):
        """norm3 -> GEGLU -> ff.net.2 (+ t2) -> proj_out (+ x_in) as ONE launch (include/upk.h upk_geglu_mlp_f16), or None
        when the shape is outside the kernel's domain / the chip would not be covered (UPGPT_MLP_FUSE).
        pw1: the "_ln" GEGLU packing, pw2: Packer.append_1x1(P F2, P) — K order [h | t2]."""
        if K.MLP_FUSE == "0" or t2.C != t2.ld:
            return None
        M, C_ = t2.M, t2.C
        if (K.MLP_FUSE == "auto" and L.concurrency() > 1 and os.environ.get("UPGPT_LANES_TUNING", "1") == "1"
                and M in TUNE_CACHE_LANES.meta.get("__unfuse_mlp_M__", ())):
            # several batches in flight: the kernel holds every CU with one register-heavy workgroup that streams both
            # weights (21 us of chip time per launch); GEGLU + (ff.net.2 o proj_out) as two launches on tiles tuned for a
            # shared chip cost less (forward in flight 1.524 -> 1.490 ms) — for the row counts the table lists
            return None
        rows = self.mlp_rows(M)
        d = L.MlpDesc()
        d.x, d.ldx, d.m, d.c, d.inner = t2.t.data_ptr(), t2.ld, M, C_, pw1.n_out
        d.w1, d.b1, d.u1 = pw1.w.data_ptr(), pw1.bias.data_ptr(), pw1.ln_colsum.data_ptr()
        d.ln_eps, d.ln_dim = 1e-5, C_
        d.w2, d.b2, d.n_out, d.n_pad = pw2.w.data_ptr(), pw2.bias.data_ptr(), pw2.n_out, pw2.n_pad
        d.residual, d.ld_res = x_in.t.data_ptr(), x_in.ld
        hw = t2.H * t2.W
        d.hw, d.rows_per_wg = hw, rows
        if not self.lib.upk_geglu_mlp_supported(self.hctx, C.byref(d)):
            return None
        if K.MLP_FUSE == "auto" and (M + rows - 1) // rows < (self.ctx.num_cus * 3) // 4:
            return None  # (every workgroup streams both weights in full: it pays only when M / rows covers the chip)
        out = Act(self.alloc(M, pw2.n_out), t2.B, t2.H, t2.W, pw2.n_out)
        d.y, d.ldy = out.t.data_ptr(), out.ld
        sws = None
        if gn_stats and hw % rows == 0 and hw // rows <= 32:
            sws = self.alloc(self.ctx.gn_stats_floats(t2.B, pw2.n_pad), dtype=torch.float32)
            d.gn_stats_ws = sws.data_ptr()
            out.gn_src = Emitter.GnProvider(sws, hw // rows, pw2.n_pad)
        fn, h, chk = self.lib.upk_geglu_mlp_f16, self.hctx, self._chk
        P.add(lambda s: chk(fn(h, C.byref(d), s)), d, t2, x_in, pw1, pw2, out, sws, cls="igemm_k1",
              label="mlp M%d C%d rows%d" % (M, C_, rows))
        P.igemm_flops += 2 * M * (2 * pw1.n_out * C_ + pw2.n_real * (pw1.n_out + C_))
        P.flops[-1] = 2 * M * (2 * pw1.n_out * C_ + pw2.n_real * (pw1.n_out + C_))
        return out

    def head_block_ok(self, x, t, heads, dp, qk, vt_ld):
        """Whether head_block takes the transformer input x (shape inside the kernel's domain, UPGPT_HBLOCK)."""
        if K.HBLOCK == "0" or (t + ".hblock.vec") not in self.pk.w or x.C % 32:
            return False
        rows = K.XB_ROWS or 32
        d = L.HblockDesc()
        d.ldx, d.m, d.c, d.heads, d.d = x.C, x.M, x.C, heads, dp
        d.ld_t0, d.ld_qk, d.vt_ld, d.hw, d.rows_per_wg = x.C, qk.ld, vt_ld, x.H * x.W, rows
        if not self.lib.upk_head_block_supported(self.hctx, C.byref(d)):
            return False
        return K.HBLOCK == "1" or x.M // rows >= self.ctx.num_cus

    def head_block(self, P, x, n, t, heads, dp, qk, vt, vt_ld, gn):
        """SpatialTransformer.norm -> proj_in -> norm1 -> q | k | v (include/upk.h upk_head_block_f16); returns t0.
        gn = (gamma, beta, eps, ws): the GroupNorm of x.  When the producer of x left per-(row block, channel) partial
        statistics (decided when the program runs, as in Emitter.groupnorm) the normalisation happens on the tile inside
        the kernel: ONE launch; otherwise a GroupNorm launch writes xn first.  Call head_block_ok first."""
        w = self.pk.w
        vec = w[t + ".hblock.vec"]
        gamma, beta, eps, ws = gn
        M, C_ = x.M, x.C
        hw = x.H * x.W
        rows = K.XB_ROWS or 32
        pi, qkv = w[n + ".proj_in"], w[t + ".attn1.qkv_ln"]
        armed = self._arm_gn_sources([x]) if K.HBLOCK_GN else None
        if armed is None:
            xn = self.groupnorm(P, x, gamma, beta, eps, False, ws)
        else:
            xn = Act(self.alloc(M, C_), x.B, x.H, x.W, C_)  # (written only when the statistics are not of the usable kind)
        d = L.HblockDesc()
        d.x, d.ldx, d.m, d.c, d.heads, d.d = xn.t.data_ptr(), xn.ld, M, C_, heads, dp
        d.w_in, d.w_qkv, d.vec = pi.w.data_ptr(), qkv.w.data_ptr(), vec.data_ptr()
        d.ln_eps, d.ln_dim = 1e-5, C_
        d.qk, d.ld_qk, d.vt, d.vt_ld = qk.t.data_ptr(), qk.ld, vt.data_ptr(), vt_ld
        d.hw, d.rows_per_wg = hw, rows
        require(self.lib.upk_head_block_supported(self.hctx, C.byref(d)), "head_block: unsupported shape", RuntimeError)
        t0 = Act(self.alloc(M, C_), x.B, x.H, x.W, C_)
        d.t0, d.ld_t0 = t0.t.data_ptr(), t0.ld
        fn, h, chk = self.lib.upk_head_block_f16, self.hctx, self._chk
        label = "hblock M%d C%d d%d rows%d" % (M, C_, dp, rows)
        if armed is None:
            P.add(lambda s: chk(fn(h, C.byref(d), s)), d, xn, pi, qkv, vec, t0, qk, vt, cls="igemm_k1", label=label)
        else:
            fused_fn, apply_fn = self.lib.upk_conv_gn_fused, self.lib.upk_groupnorm_apply_nhwc_f16
            fin_fn, gn_fn = self.lib.upk_groupnorm_finalize_f32, self.lib.upk_groupnorm_nhwc_f16
            a = (x.t.data_ptr(), x.C, x.ld, None, 0, 0, x.B, hw, 32, gamma.data_ptr(), beta.data_ptr(), float(eps), 0,
                 xn.t.data_ptr(), xn.ld)
            src, sws = armed[0]
            mine = None
            if K.GN_REDUCE_APPLY and not isinstance(src, Emitter.GnProvider) and not src.gno_y:
                # a producer that splits K normalises in its reduce pass (include/upk.h gno_*), as Emitter.groupnorm arms
                # it: the head then reads xn and runs without the in-kernel GroupNorm
                mine = src
                mine.gno_gamma, mine.gno_beta, mine.gno_eps = gamma.data_ptr(), beta.data_ptr(), float(eps)
                mine.gno_silu, mine.gno_y, mine.gno_ld, mine.gno_skip_y = 0, xn.t.data_ptr(), xn.ld, 0

            def run(s):
                if isinstance(src, Emitter.GnProvider):
                    mode, nb, ld = 2, src.nblk, src.ld
                else:
                    m_, n_ = C.c_int(0), C.c_int(0)
                    chk(fused_fn(h, C.byref(src), C.byref(m_), C.byref(n_)))
                    mode, nb, ld = (m_.value if m_.value != 3 or src is mine else 0), n_.value, src.n_pad
                if mode == 3:  # (xn was written by the producer's reduce pass)
                    d.x, d.ldx, d.gn_part = xn.t.data_ptr(), xn.ld, None
                    chk(fn(h, C.byref(d), s))
                    return
                if mode == 2 and nb <= 32:
                    d.x, d.ldx = x.t.data_ptr(), x.ld
                    d.gn_part, d.gn_gamma, d.gn_beta = sws.data_ptr(), gamma.data_ptr(), beta.data_ptr()
                    d.gn_nblk, d.gn_ld, d.gn_groups, d.gn_eps = nb, ld, 32, float(eps)
                    chk(fn(h, C.byref(d), s))
                    return
                if mode == 2:
                    chk(fin_fn(h, sws.data_ptr(), nb, ld, x.B, hw, x.C, 32, ws.data_ptr(), s))
                    chk(apply_fn(h, *a, ws.data_ptr(), 1, 0, 0, None, 0, 0, s))
                elif mode:
                    chk(apply_fn(h, *a, sws.data_ptr(), mode, nb, ld, None, 0, 0, s))
                else:
                    chk(gn_fn(h, *a, ws.data_ptr(), s))
                d.x, d.ldx, d.gn_part = xn.t.data_ptr(), xn.ld, None
                chk(fn(h, C.byref(d), s))

            P.add(run, d, x, xn, gamma, beta, ws, armed, pi, qkv, vec, t0, qk, vt, cls="igemm_k1", label=label + " gn")
        fl = 2 * M * (pi.k_real * pi.n_real + qkv.k_real * qkv.n_real)
        P.igemm_flops += fl
        P.flops[-1] = fl
        return t0

    def cross_block(self, P, a1, t0, t, kc, vtc, cld, heads, dp, scale):
        """attn1.to_out (+ t0) -> norm2 -> attn2.to_q -> attention over the context -> attn2.to_out (+ t1) as ONE launch
        (include/upk.h upk_cross_block_f16), or None when the shape is outside the kernel's domain (UPGPT_XBLOCK)."""
        w = self.pk.w
        vec = w.get(t + ".xblock.vec")
        if K.XBLOCK == "0" or vec is None or t0.C != t0.ld:
            return None
        M, C_ = t0.M, t0.C
        hw = t0.H * t0.W
        rows = K.XB_ROWS or (32 if M // 32 >= self.ctx.num_cus else 16)
        o1, o2, ql = w[t + ".attn1.to_out"], w[t + ".attn2.to_out"], w[t + ".attn2.q_ln"]
        d = L.XblockDesc()
        d.a1, d.lda, d.m, d.c, d.heads, d.d = a1.t.data_ptr(), a1.ld, M, C_, heads, dp
        d.t0, d.ld_t0 = t0.t.data_ptr(), t0.ld
        d.w_out1, d.w_q, d.w_out2, d.vec = o1.w.data_ptr(), ql.w.data_ptr(), o2.w.data_ptr(), vec.data_ptr()
        d.ln_eps, d.ln_dim = 1e-5, C_
        d.k_ctx, d.ldk, d.n_kv = kc.t.data_ptr(), kc.ld, self.n_ctx
        d.vt_ctx, d.vt_ld, d.scale = vtc.data_ptr(), cld, float(scale)
        d.hw, d.rows_per_wg = hw, rows
        if not self.lib.upk_cross_block_supported(self.hctx, C.byref(d)):
            return None
        if K.XBLOCK == "auto" and M // rows < self.ctx.num_cus:
            return None  # (every workgroup streams the three weights in full: 16x16 level 34.6 us against 32 us unfused)
        out = Act(self.alloc(M, C_), t0.B, t0.H, t0.W, C_)
        d.y, d.ldy = out.t.data_ptr(), out.ld
        fn, h, chk = self.lib.upk_cross_block_f16, self.hctx, self._chk
        # (timed with the conv / GEMM class: three of its four stages are GEMMs; its attention FLOPs are counted there too)
        P.add(lambda s: chk(fn(h, C.byref(d), s)), d, a1, t0, o1, o2, ql, vec, kc, vtc, out, cls="igemm_k1",
              label="xblock M%d C%d d%d rows%d" % (M, C_, dp, rows))
        fl = 2 * M * (o1.k_real * o1.n_real + ql.k_real * ql.n_real + o2.k_real * o2.n_real)
        fl += 4 * M * heads * self.n_ctx * (ql.n_real // heads)
        P.igemm_flops += fl
        P.flops[-1] = fl
        return out

    def attention(self, P, q, ldq, qbs, k, ldk, kbs, vt, vt_ld, out, ldo, obs, B, heads, nq, nkv, dp, scale):
        fn, h, chk = self.lib.upk_attention_f16, self.hctx, self._chk
        a = (q.data_ptr(), ldq, qbs, k.data_ptr(), ldk, kbs, vt.data_ptr(), vt_ld, out.data_ptr(), ldo, obs, B, heads,
             nq, nkv, dp, float(scale))
        P.add(lambda s: chk(fn(h, *a, s)), q, k, vt, out, cls="attention",
              label="attn B%d h%d nq%d nkv%d d%d" % (B, heads, nq, nkv, dp))



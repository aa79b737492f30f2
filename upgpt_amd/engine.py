"""Lowers arch.* records + fp32 master weights to flat programs of libupk.so launches.

Design (MI355X-first, not a translation of the reference's nn.Module.forward chain):
  * activations live as token-major NHWC fp16 [B*H*W, C] for the whole network, so
    conv1x1 == Linear == GEMM and the reference's `b c h w <-> b (h w) c` rearranges
    (attention.py:256,259) and head split/merge copies (:178,192) do not exist;
  * channel concats (openaimodel.py:736) are never materialised — GroupNorm and the
    implicit-GEMM kernel read two sources; nearest-2x upsampling is folded into the
    following conv's addressing; bias / timestep-embedding add / residual / GEGLU / SiLU
    are GEMM epilogues;
  * shapes are static per (B, H, W, n_ctx): every buffer is allocated once, every launch
    descriptor is built once, and the per-step program is captured into ONE HIP graph that
    the sampler replays S times; the step index lives on the device (no per-step host
    work, no torch.full allocations as in ddim.py:189-192);
  * step-invariant work is hoisted out of the loop: the timestep-embedding MLP and all 22
    emb_layers projections for ALL S steps (one batched GEMM each) and the cross-attention
    K / V projections of the context.

PyTorch here = device allocator + stream handle only.
"""
import ctypes as C
import json
import os

import torch

from . import _lib as L
from .arch import UNetArch, VAEArch
from ._check import require


def _rup(v, m):
    return (v + m - 1) // m * m


class TuneCache:
    """shape signature -> [cfg, splitk, best_us, default_us], measured on an MI355X by
    upk_conv_autotune and kept in-tree (upgpt_amd/tuned_gfx950.json) so that fresh processes
    start with tuned launches.  Unknown shapes fall back to the library's cost model."""

    def __init__(self, path=None):
        self.path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gfx950.json")
        self.d = {}
        self.dirty = False
        self.names = None  # configuration names the indices in the file refer to ("__configs__"), resolved by bind()
        if os.path.exists(self.path):
            try:
                with open(self.path) as f:
                    self.d = json.load(f)
            except Exception:
                self.d = {}
        self.names = self.d.pop("__configs__", None)
        self._bound = False

    def bind(self, lib):
        """Ties the stored configuration indices to THIS library's configuration list: entries are re-indexed by
        configuration name when the file records the names it was written with ("__configs__"), and entries whose
        configuration does not exist (any more) are dropped — they fall back to the cost model instead of pinning a
        different kernel or an index past the table."""
        if self._bound:
            return
        self._bound = True
        n = lib.upk_conv_num_configs()
        cur = [lib.upk_conv_config_name(i).decode() for i in range(n)]
        if self.names is not None and self.names != cur:
            idx = {nm: i for i, nm in enumerate(cur)}
            remap = {i: idx.get(nm, -1) for i, nm in enumerate(self.names)}
            for k in list(self.d):
                try:
                    c = remap.get(int(self.d[k][0]), -1)
                except (TypeError, ValueError, IndexError, KeyError):
                    c = -1  # (a malformed entry is dropped, never a reason to fail)
                if c < 0:
                    del self.d[k]
                else:
                    self.d[k][0] = c
        else:
            for k in list(self.d):
                try:
                    ok = 0 <= int(self.d[k][0]) < n
                except (TypeError, ValueError, IndexError, KeyError):
                    ok = False
                if not ok:
                    del self.d[k]
        self.names = cur

    def get(self, key):
        return self.d.get(key)

    def put(self, key, cfg, sk, best_us, dflt_us):
        # indices written now refer to THIS library's configuration list: the stored ones must have been re-indexed first
        require(self._bound, "TuneCache.put before bind(lib): stored and new configuration indices would mix", RuntimeError)
        self.d[key] = [int(cfg), int(sk), round(float(best_us), 2), round(float(dflt_us), 2)]
        self.dirty = True
        return self.d[key]

    def save(self, path=None):
        out = dict(self.d)
        if self.names is not None:
            out["__configs__"] = self.names
        with open(path or self.path, "w") as f:
            json.dump(out, f, indent=0, sort_keys=True)
        self.dirty = False


TUNE_CACHE = TuneCache(os.environ.get("UPGPT_TUNE_FILE") or None)
# GroupNorm statistics of the VAE decoder's tensors as conv by-products (gn_stats_cap + upk_groupnorm_finalize_f32):
# measured neutral (8.29 vs 8.29 ms per decode) — the statistics pass runs at 5.6 TB/s since round 2, the channel
# partials cost the 200-us convs 2-3 % and a 32-block fold per apply workgroup more than the pass it replaces — off
VAE_GN_BYPRODUCT = os.environ.get("UPGPT_VAE_GN_BYPRODUCT", "0") == "1"
UPS_PHASES = os.environ.get("UPGPT_UPS_PHASES", "1") == "1"
LN_ROWS = os.environ.get("UPGPT_LN_ROWS", "1") == "1"
QPROJ_FUSE = os.environ.get("UPGPT_QPROJ_FUSE", "1") == "1"
GN_REDUCE_APPLY = os.environ.get("UPGPT_GN_REDUCE_APPLY", "1") == "1"
# fused feed-forward tail (csrc/mlp.hip: GEGLU -> ff.net.2 o proj_out with the hidden activation in LDS): "auto" = where
# M / rows-per-workgroup covers the chip (the 32x32 level at B = 8), "0" off, "1" wherever the kernel takes the shape
MLP_FUSE = os.environ.get("UPGPT_MLP_FUSE", "auto")
MLP_ROWS = int(os.environ.get("UPGPT_MLP_ROWS", "0"))  # rows per workgroup (32 / 64; 0 = by M)
# fused cross-attention half of a transformer block (csrc/xblock.hip: attn1.to_out -> norm2 -> to_q -> attention over the
# context -> attn2.to_out, one launch instead of three / four): "auto", "0" off, "1" wherever the kernel takes the shape
XBLOCK = os.environ.get("UPGPT_XBLOCK", "auto")
XB_ROWS = int(os.environ.get("UPGPT_XB_ROWS", "0"))  # rows per workgroup (16 / 32; 0 = by M)
# fused head of a SpatialTransformer (csrc/xblock.hip hblock_kernel: proj_in -> norm1 -> q | k | v, one launch instead of two)
HBLOCK = os.environ.get("UPGPT_HBLOCK", "auto")
HBLOCK_GN = os.environ.get("UPGPT_HBLOCK_GN", "1") == "1"  # SpatialTransformer.norm applied on the tile inside that launch
# per-XCD persistent engine (csrc/xcd.hip, include/upk.h upk_xcd_run_f16): a whole SpatialTransformer as ONE launch, sample b
# on XCD b % 8, XCD-local barriers (1.04 us measured) between its ten phases.  Built, parity-green and measured in round 5
# (DESIGN.md 12): inside the replayed forward a block costs 130 us on the engine against 95 us as a launch chain — every
# phase re-stages its rows through the XCD's shared L2 and pays 2-3 L2 / HBM round trips of 1-2.5 us that a barrier cannot
# hide — so it is OFF by default: "0" off, "1" wherever the engine takes the shape (any batch: the GPU tests), "auto" =
# batches that are a multiple of 8 at feature maps of <= XCD_MAXN pixels.  UPGPT_XCD_SPLIT=1: one launch per phase.
XCD = os.environ.get("UPGPT_XCD", "0")
XCD_MAXN = int(os.environ.get("UPGPT_XCD_MAXN", "256"))
XCD_SPLIT = os.environ.get("UPGPT_XCD_SPLIT", "0") == "1"
LN_LAUNCH_US = 5.0  # what a separate LayerNorm launch costs inside the replayed forward (3.8 us of kernel + its boundary: DESIGN.md 11i / 11j)


def head_pad(d):
    for p in (32, 64, 128, 256, 512):
        if d <= p:
            return p
    raise NotImplementedError("attention head dim %d > 512" % d)


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


class PW:
    """A packed weight: fp16 tiles + fp32 bias in packed row order."""
    __slots__ = ("w", "n_pad", "bias", "n_out", "k_packed", "ksize", "n_real", "k_real", "ln_colsum", "k_append", "w_phase")


class PWX:
    """A weight packed for the per-XCD engine (Packer.pack_xcd)."""
    __slots__ = ("w", "bias", "ntiles", "k", "n", "colsum")


class Packer:
    """fp32 OIHW / [out,in] master weights -> libupk packed fp16 (done once per weight set)."""

    def __init__(self, ctx, get):
        self.ctx, self.get, self.dev = ctx, get, ctx.device

    def _maps(self, m):
        return None if m is None else torch.as_tensor(m, dtype=torch.int32, device=self.dev).contiguous()

    def pack(self, names, row_map=None, col_map=None, cin_packed=None, bias=True, n_out=None, ln=None):
        """`names`: one weight name or a list whose rows are concatenated (fused q|k|v).
        `ln`: name of a LayerNorm whose affine is folded into this Linear (include/upk.h ln_colsum):
        W' = W * gamma, bias' = bias + W @ beta, plus the column sums of the fp16-rounded W'."""
        if isinstance(names, str):
            names = [names]
        ws = [self.get(n + ".weight") for n in names]
        w = ws[0] if len(ws) == 1 else torch.cat([x.reshape(x.shape[0], -1) for x in ws], 0).reshape(
            -1, *ws[0].shape[1:])
        w = w.contiguous().float()
        ln_bias = None
        if ln is not None:
            require(w.dim() == 2 and col_map is None, "a LayerNorm can only be folded into a Linear without a column map", ValueError)
            gamma, beta = self.get(ln + ".weight").float(), self.get(ln + ".bias").float()
            ln_bias = w @ beta
            w = (w * gamma[None, :]).contiguous()
        rm, cm = self._maps(row_map), self._maps(col_map)
        p = PW()
        p.w, p.n_pad = self.ctx.pack_weight(w, row_map=rm, col_map=cm, cin_packed=cin_packed)
        p.ksize = w.shape[-1] if w.dim() == 4 else 1
        cin = w.shape[1]
        p.k_packed = (cin_packed if cin_packed is not None else _rup(cin if cm is None else cm.numel(), 32))
        n_rows = w.shape[0] if rm is None else rm.numel()
        p.n_out = n_rows if n_out is None else n_out
        p.n_real = w.shape[0] if rm is None else int((rm >= 0).sum().item())
        p.k_real = (cin if cm is None else int((cm >= 0).sum().item())) * p.ksize * p.ksize
        def rows_packed(vec):  # per-output-row vector -> packed row order, zero padded to n_pad
            out = torch.zeros(p.n_pad, dtype=torch.float32, device=self.dev)
            if rm is None:
                out[: vec.numel()] = vec
            else:
                idx = rm.long()
                out[: idx.numel()] = torch.where(idx >= 0, vec[idx.clamp(min=0)], torch.zeros((), device=self.dev))
            return out

        p.bias = None
        p.ln_colsum = None
        p.k_append = 0
        p.w_phase = None
        b = None
        if bias:
            bs = [self.get(n + ".bias") for n in names]
            b = (bs[0] if len(bs) == 1 else torch.cat(bs, 0)).float()
        if ln_bias is not None:
            b = ln_bias if b is None else b + ln_bias
            p.ln_colsum = rows_packed(w.half().float().sum(dim=1))
        if b is not None:
            p.bias = rows_packed(b)
        return p

    def add_upsample_phases(self, p, name):
        """Phase weights of an Upsample conv (include/upk.h w_phase): nearest 2x + conv3x3 = four 2x2 convs on the
        low-resolution grid; tap (ty, tx) of phase (py, px) = sum of the 3x3 taps that read the same low-resolution
        pixel (summed in fp32, rounded to fp16 once)."""
        w = self.get(name + ".weight").float().to(self.dev)
        require(w.dim() == 4 and w.shape[-1] == 3 and w.shape[-2] == 3, "upsample phase weights need a 3x3 conv weight", ValueError)
        taps = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}  # (phase bit, tap) -> 3x3 taps
        parts = []
        for py in (0, 1):
            for px in (0, 1):
                wp = torch.zeros(w.shape[0], w.shape[1], 2, 2, device=self.dev)
                for ty in (0, 1):
                    for tx in (0, 1):
                        for ky in taps[(py, ty)]:
                            for kx in taps[(px, tx)]:
                                wp[:, :, ty, tx] += w[:, :, ky, kx]
                packed, n_pad = self.ctx.pack_weight(wp.contiguous())
                require(n_pad == p.n_pad, "phase weight rows differ from the 3x3 weight's", RuntimeError)
                parts.append(packed.reshape(-1))
        p.w_phase = torch.cat(parts).contiguous()
        return p

    def append_1x1(self, main, skip):
        """`main` followed along K by the 1x1 weight `skip` (include/upk.h: appended K segment): one launch computes
        conv(main) + conv1x1(skip) — a ResBlock's second conv plus its skip projection (openaimodel.py:274-275)."""
        require(skip.ksize == 1 and skip.n_pad == main.n_pad and skip.n_out == main.n_out and main.ln_colsum is None, "append_1x1: the appended weight must be a 1x1 with the main weight's rows", ValueError)
        p = PW()
        p.w = torch.cat([main.w.reshape(-1), skip.w.reshape(-1)])
        p.n_pad, p.n_out, p.ksize, p.k_packed, p.n_real = main.n_pad, main.n_out, main.ksize, main.k_packed, main.n_real
        p.k_append = skip.k_packed
        p.k_real = main.k_real + skip.k_real
        p.ln_colsum = None
        bs = [b for b in (main.bias, skip.bias) if b is not None]
        p.bias = None if not bs else (bs[0] if len(bs) == 1 else bs[0] + bs[1])
        return p

    def pack_product(self, outer, inner):
        """The Linear `inner` followed by the Linear / 1x1 conv `outer` with nothing in between, as ONE weight:
        W = W_outer @ W_inner (fp32, then packed fp16), bias = W_outer @ b_inner (`outer`'s own bias is left to the
        caller: Packer.append_1x1 adds it with the appended segment)."""
        wo = self.get(outer + ".weight").float()
        wo = wo.reshape(wo.shape[0], -1)
        wi = self.get(inner + ".weight").float()
        prod = {"p.weight": (wo @ wi).contiguous(), "p.bias": wo @ self.get(inner + ".bias").float()}
        return Packer(self.ctx, lambda n: prod[n]).pack("p")

    def vec(self, name):
        return self.get(name).float().contiguous()

    def pack_xcd(self, w, bias=None, rows=None, cols=None):
        """[N, K] fp32 -> operands of a upk_xphase GEMM (include/upk.h): fp16 tiles [N/16][K/32][64 lanes][8] (lane
        16 g + i holds W[16 t + i][32 kc + 8 g .. + 7]) and the fp32 bias in tile order, N padded to 16 and K to 32 with
        zeros.  rows / cols: index tensors (packed row / column <- source row / column, -1 = zero), applied first."""
        w = w.float().to(self.dev)
        if rows is not None:
            r = torch.as_tensor(rows, device=self.dev).long()
            wr = w.new_zeros(r.numel(), w.shape[1])
            wr[r >= 0] = w[r[r >= 0]]
            if bias is not None:
                br = w.new_zeros(r.numel())
                br[r >= 0] = bias.float().to(self.dev)[r[r >= 0]]
                bias = br
            w = wr
        if cols is not None:
            c = torch.as_tensor(cols, device=self.dev).long()
            wc = w.new_zeros(w.shape[0], c.numel())
            wc[:, c >= 0] = w[:, c[c >= 0]]
            w = wc
        N, K = w.shape
        N16, K32 = _rup(N, 16), _rup(K, 32)
        wp = w.new_zeros(N16, K32)
        wp[:N, :K] = w
        T, KC = N16 // 16, K32 // 32
        px = PWX()
        px.w = wp.half().view(T, 16, KC, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)
        px.bias = None
        if bias is not None:
            px.bias = w.new_zeros(N16)
            px.bias[:N] = bias.float().to(self.dev)
        px.ntiles, px.k, px.n = T, K32, N
        px.colsum = wp.half().float().sum(dim=1).contiguous()  # (of the fp16-rounded rows: folded-LayerNorm GEMMs)
        return px


def qproj_pack(w, gamma, beta, heads, dh, dp, cq, dev):
    """to_q weight [heads*dh, C] (+ the LayerNorm affine in front of it) -> operands of upk_attention_qproj_f16
    (include/upk.h): fp16 [heads][dp (row-permuted)][cq] with gamma folded in, fp32 column sums of the rounded rows and
    W beta, both [heads*dp] in natural order (head dim zero-padded dh -> dp, channels C -> cq)."""
    C_ = w.shape[1]
    wf = torch.zeros(heads, dp, cq, dtype=torch.float32, device=dev)
    wf[:, :dh, :C_] = (w.float().to(dev) * gamma.float().to(dev)[None, :]).reshape(heads, dh, C_)
    bias = torch.zeros(heads, dp, dtype=torch.float32, device=dev)
    bias[:, :dh] = (w.float().to(dev) @ beta.float().to(dev)).reshape(heads, dh)
    w16 = wf.half()
    colsum = w16.float().sum(dim=2)
    r = torch.arange(dp, device=dev)
    kd, t, m = r // 32, (r % 32) // 16, r % 16
    src = 32 * kd + 8 * (m // 4) + 4 * t + (m % 4)  # packed row r holds natural row src
    return w16[:, src, :].contiguous(), colsum.reshape(-1).contiguous(), bias.reshape(-1).contiguous()


def pad_rows_map(parts, heads, dh, dp):
    """Row map of `parts` stacked [heads*dh]-row matrices -> [heads*dp]-row blocks each
    (head dim zero-padded dh -> dp)."""
    j = torch.arange(parts * heads * dp)
    part, r = j // (heads * dp), j % (heads * dp)
    h, d = r // dp, r % dp
    return torch.where(d < dh, part * heads * dh + h * dh + d, torch.full_like(j, -1))


def geglu_rows_map(inner):
    """Per 64 packed rows: [32 value rows | 32 gate rows] (include/upk.h UPK_F_GEGLU)."""
    j = torch.arange(2 * inner)
    blk, r = j // 64, j % 64
    return torch.where(r < 32, blk * 32 + r, inner + blk * 32 + (r - 32))


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
        cache = TUNE_CACHE if cache is None else cache
        cache.bind(self.lib)
        hits = tuned = missing = 0
        for d, key in self.convs:
            ent = cache.get(key)
            if ent is None and not tune_missing and key.endswith("_gs"):
                ent = cache.get(key[:-3])  # (statistics by-product armed on a shape that was tuned without it)
            if ent is None and not tune_missing and not key.endswith("_gs"):
                ent = cache.get(key + "_gs")  # (tuned with the statistics by-product armed; the choice is valid without)
            if ent is None and not tune_missing and key.endswith("_lnr"):
                ent = cache.get(key[:-4])  # (tuned as a plain GEMM; usable only if that choice does not split K)
                if ent is not None and int(ent[1]) != 1 and not self._is_as(int(ent[0])):
                    ent = None
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
        return hits, tuned, missing

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
        if LN_ROWS and prod is not None and not prod.ln_rows_out and x.C == x.ld:
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
                fold = e_ln[2] < e_pl[2] + LN_LAUNCH_US
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
        d.n_out, d.n_pad = pw.n_out, pw.n_pad
        if pw.bias is not None:
            d.bias = pw.bias.data_ptr()
        phased = UPS_PHASES and bool(flags & L.F_UPSAMPLE2X) and pw.w_phase is not None and x2 is None
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
            if GN_REDUCE_APPLY and x2 is None and not isinstance(armed[0][0], Emitter.GnProvider) and not armed[0][0].gno_y:
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
        if MLP_ROWS in (32, 64):
            return MLP_ROWS
        return 64 if M // 64 >= self.ctx.num_cus else 32

    def geglu_mlp(self, P, t2, x_in, pw1, pw2, gn_stats=True):
        """norm3 -> GEGLU -> ff.net.2 (+ t2) -> proj_out (+ x_in) as ONE launch (include/upk.h upk_geglu_mlp_f16), or None
        when the shape is outside the kernel's domain / the chip would not be covered (UPGPT_MLP_FUSE).
        pw1: the "_ln" GEGLU packing, pw2: Packer.append_1x1(P F2, P) — K order [h | t2]."""
        if MLP_FUSE == "0" or t2.C != t2.ld:
            return None
        M, C_ = t2.M, t2.C
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
        if MLP_FUSE == "auto" and (M + rows - 1) // rows < (self.ctx.num_cus * 3) // 4:
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
        if HBLOCK == "0" or (t + ".hblock.vec") not in self.pk.w or x.C % 32:
            return False
        rows = XB_ROWS or 32
        d = L.HblockDesc()
        d.ldx, d.m, d.c, d.heads, d.d = x.C, x.M, x.C, heads, dp
        d.ld_t0, d.ld_qk, d.vt_ld, d.hw, d.rows_per_wg = x.C, qk.ld, vt_ld, x.H * x.W, rows
        if not self.lib.upk_head_block_supported(self.hctx, C.byref(d)):
            return False
        return HBLOCK == "1" or x.M // rows >= self.ctx.num_cus

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
        rows = XB_ROWS or 32
        pi, qkv = w[n + ".proj_in"], w[t + ".attn1.qkv_ln"]
        armed = self._arm_gn_sources([x]) if HBLOCK_GN else None
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
            if GN_REDUCE_APPLY and not isinstance(src, Emitter.GnProvider) and not src.gno_y:
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
        if XBLOCK == "0" or vec is None or t0.C != t0.ld:
            return None
        M, C_ = t0.M, t0.C
        hw = t0.H * t0.W
        rows = XB_ROWS or (32 if M // 32 >= self.ctx.num_cus else 16)
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
        if XBLOCK == "auto" and M // rows < self.ctx.num_cus:
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


# ====================================================================== UNet
class PackedUNet:
    """All UNet weights packed for the kernels (independent of batch / resolution)."""

    def __init__(self, ctx, arch: UNetArch, get):
        pk = Packer(ctx, get)
        self.arch = arch
        mc, te = arch.model_channels, arch.time_embed_dim
        if mc % 32:
            raise NotImplementedError("model_channels must be a multiple of 32 (got %d)" % mc)
        w = {}
        w["time_embed.0"] = pk.pack("time_embed.0")
        w["time_embed.2"] = pk.pack("time_embed.2")
        v = {}

        def norm(name):
            v[name] = (pk.vec(name + ".weight"), pk.vec(name + ".bias"))

        for Lr in arch.all_layers():
            n = Lr.name
            if Lr.kind == "conv":
                w[n] = pk.pack(n, cin_packed=_rup(Lr.cin, 32))
            elif Lr.kind == "res":
                norm(n + ".in_layers.0")
                w[n + ".in_layers.2"] = pk.pack(n + ".in_layers.2")
                w[n + ".emb_layers.1"] = pk.pack(n + ".emb_layers.1")
                norm(n + ".out_layers.0")
                w[n + ".out_layers.3"] = pk.pack(n + ".out_layers.3")
                if Lr.cin != Lr.cout:
                    w[n + ".skip_connection"] = pk.pack(n + ".skip_connection")
                    w[n + ".out_layers.3+skip"] = pk.append_1x1(w[n + ".out_layers.3"], w[n + ".skip_connection"])
            elif Lr.kind == "st":
                if Lr.depth != 1:
                    raise NotImplementedError("transformer_depth != 1")
                heads, dh = Lr.heads, Lr.dhead
                dp = head_pad(dh)
                hd = heads * dp
                norm(n + ".norm")
                w[n + ".proj_in"] = pk.pack(n + ".proj_in")
                t = n + ".transformer_blocks.0"
                to_out_cols = pad_rows_map(1, heads, dh, dp)
                # norm1 / norm2 / norm3 can be folded into their only consumers (Emitter.ln_linear decides
                # per shape): both packings are kept, "<name>_ln" has the LayerNorm affine folded in
                for sfx, fold in (("", None), ("_ln", True)):
                    w[t + ".attn1.qkv" + sfx] = pk.pack(
                        [t + ".attn1.to_q", t + ".attn1.to_k", t + ".attn1.to_v"], row_map=pad_rows_map(3, heads, dh, dp),
                        bias=False, n_out=2 * hd, ln=(t + ".norm1") if fold else None)
                    w[t + ".attn2.q" + sfx] = pk.pack(t + ".attn2.to_q", row_map=pad_rows_map(1, heads, dh, dp),
                                                      bias=False, ln=(t + ".norm2") if fold else None)
                    w[t + ".ff.geglu" + sfx] = pk.pack(t + ".ff.net.0.proj", row_map=geglu_rows_map(4 * heads * dh),
                                                       n_out=4 * heads * dh, ln=(t + ".norm3") if fold else None)
                if dp in (32, 64) and Lr.ch % 224 == 0:  # operands of the attention that projects its own queries
                    w[t + ".attn2.qproj"] = qproj_pack(get(t + ".attn2.to_q.weight"), get(t + ".norm2.weight"),
                                                       get(t + ".norm2.bias"), heads, dh, dp, Lr.ch, ctx.device)
                w[t + ".attn1.to_out"] = pk.pack(t + ".attn1.to_out.0", col_map=to_out_cols)
                w[t + ".attn2.kv"] = pk.pack([t + ".attn2.to_k", t + ".attn2.to_v"],
                                             row_map=pad_rows_map(2, heads, dh, dp), bias=False, n_out=hd)
                w[t + ".attn2.to_out"] = pk.pack(t + ".attn2.to_out.0", col_map=to_out_cols)
                # epilogue vectors of the fused head (include/upk.h upk_hblock_desc.vec)
                pi, qkv = w[n + ".proj_in"], w[t + ".attn1.qkv_ln"]
                if pi.n_pad == Lr.ch and qkv.n_pad == 3 * hd and pi.ksize == 1:
                    vec = torch.cat([pi.bias, qkv.ln_colsum, qkv.bias])
                    w[t + ".hblock.vec"] = torch.cat([vec, vec.new_zeros(-vec.numel() % 256)]).contiguous()
                # epilogue vectors of the fused cross-attention half (include/upk.h upk_xblock_desc.vec)
                o1, o2, ql = w[t + ".attn1.to_out"], w[t + ".attn2.to_out"], w[t + ".attn2.q_ln"]
                if o1.n_pad == Lr.ch and o2.n_pad == Lr.ch and ql.n_pad == hd:
                    vec = torch.cat([o1.bias, ql.ln_colsum, ql.bias, o2.bias])
                    w[t + ".xblock.vec"] = torch.cat([vec, vec.new_zeros(-vec.numel() % 256)]).contiguous()
                inner = heads * dh
                w[t + ".ff.out"] = pk.pack(t + ".ff.net.2")
                for k in ("norm1", "norm2", "norm3"):
                    norm(t + "." + k)
                w[n + ".proj_out"] = pk.pack(n + ".proj_out")
                # proj_out(t2 + ff.net.2(h)) = (P F2) h + P t2 + (P b2 + bp): the block's last Linear and the
                # transformer's output projection as one GEMM with t2 as an appended K segment (Emitter.fold_ff_out)
                w[n + ".ff.out+proj_out"] = pk.append_1x1(pk.pack_product(n + ".proj_out", t + ".ff.net.2"),
                                                          w[n + ".proj_out"])
                if XCD != "0" and 32 % heads == 0 and dp in (32, 64, 128) and Lr.ch % 32 == 0:
                    w[n + ".xcd"] = self._pack_xcd_block(pk, get, n, t, Lr, heads, dh, dp)
            elif Lr.kind == "down":
                w[n + ".op"] = pk.pack(n + ".op")
            elif Lr.kind == "up":
                w[n + ".conv"] = pk.add_upsample_phases(pk.pack(n + ".conv"), n + ".conv")
        norm("out.0")
        w["out.2"] = pk.pack("out.2")
        self.w, self.v = w, v


def _pack_xcd_block(self, pk, get, n, t, Lr, heads, dh, dp):
    """The seven GEMMs of a SpatialTransformer as per-XCD engine operands (include/upk.h upk_xphase).  Every norm in front
    of a Linear is folded into it: W' = W * gamma (per input column), b' = b + W beta — SpatialTransformer.norm
    (attention.py:254) into proj_in, norm1 / norm2 / norm3 (attention.py:212-215) into q|k|v, attn2.to_q and the GEGLU
    projection; the engine's GroupNorm / LayerNorm then only subtract the mean and scale by rstd."""
    f = lambda name: get(name).float().to(pk.dev)
    C_ = Lr.ch
    inner = 4 * heads * dh

    def folded(wname, gname, bias=None):
        W = f(wname + ".weight")
        W = W.reshape(W.shape[0], -1)
        b = W @ f(gname + ".bias")
        if bias is not None:
            b = b + f(bias)
        return W * f(gname + ".weight")[None, :], b

    o = {}
    Wi, bi = folded(n + ".proj_in", n + ".norm", n + ".proj_in.bias")
    o["proj_in"] = pk.pack_xcd(Wi, bi)
    Wq = torch.cat([f(t + ".attn1.to_q.weight"), f(t + ".attn1.to_k.weight"), f(t + ".attn1.to_v.weight")], 0)
    g1, b1 = f(t + ".norm1.weight"), f(t + ".norm1.bias")
    o["qkv"] = pk.pack_xcd(Wq * g1[None, :], Wq @ b1, rows=pad_rows_map(3, heads, dh, dp))
    hcols = pad_rows_map(1, heads, dh, dp)
    o["out1"] = pk.pack_xcd(f(t + ".attn1.to_out.0.weight"), f(t + ".attn1.to_out.0.bias"), cols=hcols)
    W2, b2 = folded(t + ".attn2.to_q", t + ".norm2")
    o["q2"] = pk.pack_xcd(W2, b2, rows=hcols)
    o["out2"] = pk.pack_xcd(f(t + ".attn2.to_out.0.weight"), f(t + ".attn2.to_out.0.bias"), cols=hcols)
    Wg, bg = folded(t + ".ff.net.0.proj", t + ".norm3", t + ".ff.net.0.proj.bias")
    u = torch.arange(2 * inner)
    tile, i = u // 16, u % 16
    grows = torch.where(tile % 2 == 0, (tile // 2) * 16 + i, inner + (tile // 2) * 16 + i)  # tiles alternate value / gate
    o["geglu"] = pk.pack_xcd(Wg, bg, rows=grows)
    # proj_out(t2 + ff.net.2(h)) + x = (P F2) h + P t2 + (P b2 + bp) + x: one GEMM over [h | t2]
    Pw = f(n + ".proj_out.weight")
    Pw = Pw.reshape(Pw.shape[0], -1)
    F2, c2 = f(t + ".ff.net.2.weight"), f(t + ".ff.net.2.bias")
    o["ffout"] = pk.pack_xcd(torch.cat([Pw @ F2, Pw], 1), Pw @ c2 + f(n + ".proj_out.bias"))
    o["inner"] = inner
    return o


PackedUNet._pack_xcd_block = _pack_xcd_block


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


class UNetPlan(Emitter):
    """Buffers + programs of one UNet for fixed (B, H, W, n_ctx, rows).

    rows = number of timestep-embedding rows: B in 'forward' mode (per-sample t, as
    UNetModel.forward takes them), S in 'sampler' mode (one row per DDIM step, shared by
    the batch: ddim.py:142 uses the same t for every sample)."""

    def __init__(self, ctx, packed: PackedUNet, B, H, W, n_ctx, rows, mode):
        super().__init__(ctx)
        require(mode in ("forward", "sampler"), "UNetPlan mode must be 'forward' or 'sampler'", ValueError)
        self.pk, self.arch = packed, packed.arch
        self.B, self.H, self.W, self.n_ctx, self.rows, self.mode = B, H, W, n_ctx, rows, mode
        a = self.arch
        mc, te = a.model_channels, a.time_embed_dim
        self.cin_pad = _rup(a.in_channels, 32)
        # inputs / outputs
        self.xin = Act(self.alloc(B * H * W, self.cin_pad, zero=True), B, H, W, a.in_channels)
        self.ctx32 = self.alloc(B * n_ctx, a.context_dim, dtype=torch.float32)
        self.ctx16 = Act(self.alloc(B * n_ctx, a.context_dim), 1, B * n_ctx, 1, a.context_dim)
        self.t_rows = self.alloc(rows, dtype=torch.float32)
        self.eps = self.alloc(B, a.out_channels, H, W, dtype=torch.float32)
        self.step = self.alloc(1, dtype=torch.int32, zero=True)
        self.gn_ws = self.alloc(max(64, ctx.groupnorm_ws_bytes(B, H * W) // 4), dtype=torch.float32)
        self.rowvecs = {}
        self.kv = {}
        self.prep = Program(ctx)
        self.body = Program(ctx)
        self._emit_prep()
        self._emit_body()

    # ---- step-invariant part
    def _emit_prep(self):
        P, a, w = self.prep, self.arch, self.pk.w
        mc, te, R = a.model_channels, a.time_embed_dim, self.rows
        lib, h, chk = self.lib, self.hctx, self._chk
        c32, c16 = self.ctx32, self.ctx16
        P.add(lambda s: chk(lib.upk_f32_to_f16(h, c32.data_ptr(), c32.shape[0], c32.shape[1], c16.t.data_ptr(),
                                               c16.ld, s)))
        temb = Act(self.alloc(R, mc), 1, R, 1, mc)
        tr = self.t_rows
        P.add(lambda s: chk(lib.upk_timestep_embed_f16(h, tr.data_ptr(), R, mc, 10000.0, temb.t.data_ptr(), mc, s)))
        # emb = Linear(SiLU(Linear(temb))) (openaimodel.py:506-511); every consumer applies SiLU
        # first (openaimodel.py:219), so SiLU(emb) is what is kept.
        h1 = self.conv(P, temb, w["time_embed.0"], flags=L.F_SILU)
        semb = self.conv(P, h1, w["time_embed.2"], flags=L.F_SILU)
        for Lr in a.all_layers():
            if Lr.kind == "res":
                rv = self.alloc(R, Lr.cout, dtype=torch.float32)
                self.conv(P, semb, w[Lr.name + ".emb_layers.1"], out_f32=rv)
                self.rowvecs[Lr.name] = rv
            elif Lr.kind == "st":
                heads, dp = Lr.heads, head_pad(Lr.dhead)
                hd = heads * dp
                vt_ld = _rup(self.n_ctx, 32)
                kc = Act(self.alloc(self.B * self.n_ctx, hd), 1, self.B * self.n_ctx, 1, hd)
                vtc = self.alloc(self.B, heads, dp, vt_ld, zero=True)
                self.conv(P, self.ctx16, w[Lr.name + ".transformer_blocks.0.attn2.kv"], out=kc,
                          vt=dict(t=vtc, heads=heads, dhead=dp, ld=vt_ld, tokens=self.n_ctx, **{"from": hd}))
                self.kv[Lr.name] = (kc, vtc, vt_ld)

    # ---- per-step part
    def _rv(self, name):
        rv = self.rowvecs[name]
        if self.mode == "sampler":
            return dict(rowvec=rv, rv_bs=0, rv_ss=rv.shape[1], step=self.step)
        return dict(rowvec=rv, rv_bs=rv.shape[1], rv_ss=0, step=None)

    def _res(self, P, Lr, x, skip):
        w, v = self.pk.w, self.pk.v
        n = Lr.name
        if skip is not None and (skip.H, skip.W, skip.B) != (x.H, x.W, x.B):
            # the reference fails in torch.cat here (openaimodel.py:736) when H or W is not a
            # multiple of 2^(levels-1); fail just as loudly instead of reading mismatched rows
            raise ValueError("UNet skip connection %dx%d does not match the decoder feature map %dx%d at %s: "
                             "latent height and width must be multiples of %d" % (
                                 skip.H, skip.W, x.H, x.W, n, 2 ** (len(self.arch.channel_mult) - 1)))
        g1, b1 = v[n + ".in_layers.0"]
        hh = self.conv(P, x, w[n + ".in_layers.2"], x2=skip, gn=(g1, b1, 1e-5, True, self.gn_ws), gn_stats=True,
                       **self._rv(n))
        gn2 = (*v[n + ".out_layers.0"], 1e-5, True, self.gn_ws, True)  # (hh has no reader but this GroupNorm)
        if Lr.cin != Lr.cout:
            if self.fold_skip(hh, w[n + ".out_layers.3"], w[n + ".skip_connection"], x, skip):
                # skip projection as an appended K segment of the second conv: one launch, no residual round trip
                return self.conv(P, hh, w[n + ".out_layers.3+skip"], append=(x, skip), gn=gn2, gn_stats=True)
            sk = self.conv(P, x, w[n + ".skip_connection"], x2=skip)
        else:
            require(skip is None, "decoder ResBlock without channel change got a skip tensor", RuntimeError)
            sk = x
        return self.conv(P, hh, w[n + ".out_layers.3"], residual=sk, gn=gn2, gn_stats=True)

    def xcd_block(self, P, Lr, x):
        """The whole SpatialTransformer (attention.py:250-261) as ONE launch of the per-XCD engine (include/upk.h
        upk_xcd_run_f16; csrc/xcd.hip): GroupNorm -> proj_in -> [LN1 -> q|k|v -> self-attention -> to_out + t0] ->
        [LN2 -> to_q -> attention over the precomputed context K / V -> to_out + t1] -> [LN3 -> GEGLU] ->
        ff.net.2 o proj_out + x: ten phases, XCD-local barriers between them.  Returns the output Act, or None when the
        engine is off / does not take the shape (the caller then emits the launch chain)."""
        n = Lr.name
        wx = self.pk.w.get(n + ".xcd")
        if XCD == "0" or wx is None or n not in self.kv:
            return None
        B, HW, M, C_ = x.B, x.H * x.W, x.M, x.C
        if XCD != "1" and (B % 8 or HW > XCD_MAXN):
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
        if XCD_SPLIT:
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

    def _st(self, P, Lr, x):
        w, v = self.pk.w, self.pk.v
        n = Lr.name
        out = self.xcd_block(P, Lr, x)
        if out is not None:
            return out
        t = n + ".transformer_blocks.0"
        B, HW, M = x.B, x.H * x.W, x.M
        heads, dh = Lr.heads, Lr.dhead
        dp = head_pad(dh)
        hd = heads * dp
        scale = dh ** -0.5
        # self-attention
        qk = Act(self.alloc(M, 2 * hd), B, x.H, x.W, 2 * hd)
        vt_ld = _rup(HW, 32)
        vt = self.alloc(B, heads, dp, vt_ld, zero=True)
        t0 = None
        if self.head_block_ok(x, t, heads, dp, qk, vt_ld):
            t0 = self.head_block(P, x, n, t, heads, dp, qk, vt, vt_ld, (*v[n + ".norm"], 1e-6, self.gn_ws))
        if t0 is None:
            t0 = self.conv(P, x, w[n + ".proj_in"], gn=(*v[n + ".norm"], 1e-6, False, self.gn_ws))
            self.ln_linear(P, t0, t + ".attn1.qkv", t + ".norm1", out=qk,  # norm1 -> q|k|v (attention.py:203,212)
                           vt=dict(t=vt, heads=heads, dhead=dp, ld=vt_ld, tokens=HW, **{"from": 2 * hd}))
        a1 = Act(self.alloc(M, hd), B, x.H, x.W, hd)
        self.attention(P, qk.t, 2 * hd, HW * 2 * hd, qk.t[:, hd:], 2 * hd, HW * 2 * hd, vt, vt_ld, a1.t, hd, HW * hd,
                       B, heads, HW, HW, dp, scale)
        P.attn_flops += 4 * B * heads * HW * HW * dh
        # cross-attention over the (precomputed) context K / V
        kc, vtc, cld = self.kv[n]
        t2 = self.cross_block(P, a1, t0, t, kc, vtc, cld, heads, dp, scale)
        if t2 is not None:
            return self._st_ff(P, Lr, x, t2)
        t1 = self.conv(P, a1, w[t + ".attn1.to_out"], residual=t0)
        a2 = Act(self.alloc(M, hd), B, x.H, x.W, hd)
        # (pays while there are >= 2 waves per SIMD to hide a wave's serial projection -> scores chain: 32x32 level
        # 16 -> 12 us per block; at 16x16 (1 wave per SIMD, 4x the weight slice per wave) 16 -> 18 us)
        if (QPROJ_FUSE and (t + ".attn2.qproj") in w and t1.C == t1.ld and dp == 32
                and B * heads * ((HW + 31) // 32) >= 2048):
            # norm2 -> to_q inside the attention kernel (include/upk.h upk_attention_qproj_f16): no q GEMM, no q tensor
            wq, wu, wb = w[t + ".attn2.qproj"]
            fn, hh, chk = self.lib.upk_attention_qproj_f16, self.hctx, self._chk
            qa = (t1.t.data_ptr(), t1.ld, HW * t1.ld, t1.C, t1.C, 1e-5, wq.data_ptr(), wu.data_ptr(), wb.data_ptr(),
                  kc.t.data_ptr(), hd, self.n_ctx * hd, vtc.data_ptr(), cld, a2.t.data_ptr(), hd, HW * hd, B, heads, HW,
                  self.n_ctx, dp, float(scale))
            P.add(lambda s: chk(fn(hh, *qa, s)), t1, wq, wu, wb, kc, vtc, a2, cls="attention",
                  label="attn+q B%d h%d nq%d nkv%d d%d C%d" % (B, heads, HW, self.n_ctx, dp, t1.C))
            P.igemm_flops += 2 * M * heads * dh * t1.C
        else:
            q2 = self.ln_linear(P, t1, t + ".attn2.q", t + ".norm2")
            self.attention(P, q2.t, hd, HW * hd, kc.t, hd, self.n_ctx * hd, vtc, cld, a2.t, hd, HW * hd, B, heads, HW,
                           self.n_ctx, dp, scale)
        P.attn_flops += 4 * B * heads * HW * self.n_ctx * dh
        t2 = self.conv(P, a2, w[t + ".attn2.to_out"], residual=t1)
        return self._st_ff(P, Lr, x, t2)

    def _st_ff(self, P, Lr, x, t2):
        """GEGLU feed-forward of the block + the transformer's proj_out (attention.py:261, 330-336)."""
        w = self.pk.w
        n = Lr.name
        t = n + ".transformer_blocks.0"
        fused = self.geglu_mlp(P, t2, x, w[t + ".ff.geglu_ln"], w[n + ".ff.out+proj_out"])
        if fused is not None:
            return fused
        ff = self.ln_linear(P, t2, t + ".ff.geglu", t + ".norm3", flags=L.F_GEGLU)
        if self.fold_ff_out(ff, t2, w[t + ".ff.out"]):
            return self.conv(P, ff, w[n + ".ff.out+proj_out"], residual=x, append=(t2, None), gn_stats=True)
        t3 = self.conv(P, ff, w[t + ".ff.out"], residual=t2)
        return self.conv(P, t3, w[n + ".proj_out"], residual=x, gn_stats=True)

    def _layers(self, P, layers, x, skip=None):
        w = self.pk.w
        for Lr in layers:
            if Lr.kind == "conv":
                x = self.conv(P, x, w[Lr.name], gn_stats=True)
            elif Lr.kind == "res":
                x = self._res(P, Lr, x, skip)
                skip = None
            elif Lr.kind == "st":
                x = self._st(P, Lr, x)
            elif Lr.kind == "down":
                x = self.conv(P, x, w[Lr.name + ".op"], stride=2, gn_stats=True)
            elif Lr.kind == "up":
                x = self.conv(P, x, w[Lr.name + ".conv"], flags=L.F_UPSAMPLE2X, gn_stats=True)
        return x

    def _emit_body(self):
        P, a = self.body, self.arch
        hs = []
        x = self.xin
        self.taps = {}  # block name -> Act (debug / parity tests)
        for i, blk in enumerate(a.input_blocks):
            x = self._layers(P, blk, x)
            hs.append(x)
            self.taps["input_blocks.%d" % i] = x
        x = self._layers(P, a.middle_block, x)
        self.taps["middle_block"] = x
        for i, blk in enumerate(a.output_blocks):
            x = self._layers(P, blk, x, skip=hs.pop())
            self.taps["output_blocks.%d" % i] = x
        self.conv(P, x, self.pk.w["out.2"], nchw_out=self.eps, gn=(*self.pk.v["out.0"], 1e-5, True, self.gn_ws))

    # ---- host-facing helpers
    def close(self):
        """Releases the graphs of the sampler states attached to this plan (ddim.py / plms.py fast paths)."""
        for attr in ("_sampler_state", "_sampler_state_cfg", "_plms_state", "_plms_state_cfg"):
            st = self.__dict__.pop(attr, None)
            if st is not None:
                st.close()

    def load_context(self, context):
        """context: [B, n_ctx, context_dim] tensor (any float dtype / device)."""
        require(tuple(context.shape) == (self.B, self.n_ctx, self.arch.context_dim), lambda: "context shape %s != plan %s" % (tuple(context.shape), (self.B, self.n_ctx, self.arch.context_dim)), ValueError)
        self.ctx32.copy_(context.reshape(self.B * self.n_ctx, -1).to(self.dev, torch.float32), non_blocking=True)

    def load_x_nchw(self, x, c_off=0, zero_pad_to=0):
        """fp32 NCHW [B, c, H, W] -> channels [c_off, c_off + c) of the stem input."""
        x = x.to(self.dev, torch.float32).contiguous()
        require(x.shape[0] == self.B and tuple(x.shape[2:]) == (self.H, self.W), lambda: "stem input %s does not match the plan (B=%d, %dx%d)" % (tuple(x.shape), self.B, self.H, self.W), ValueError)
        self.ctx.nchw_to_nhwc(x, self.B, x.shape[1], self.H * self.W, self.xin.t, self.xin.ld, c_off, zero_pad_to, 1.0)


# ====================================================================== VAE decoder
class PackedVAEDecoder:
    def __init__(self, ctx, arch: VAEArch, get):
        pk = Packer(ctx, get)
        self.arch = arch
        w, v = {}, {}

        def norm(name):
            v[name] = (pk.vec("decoder." + name + ".weight"), pk.vec("decoder." + name + ".bias"))

        w["post_quant_conv"] = pk.pack("post_quant_conv", cin_packed=_rup(arch.embed_dim, 32))
        for Lr in arch.decoder:
            n, d = Lr.name, "decoder." + Lr.name
            if Lr.kind in ("conv", "conv_out"):
                w[n] = pk.pack(d, cin_packed=_rup(Lr.cin, 32))
            elif Lr.kind == "resnet":
                norm(n + ".norm1")
                w[n + ".conv1"] = pk.pack(d + ".conv1")
                norm(n + ".norm2")
                w[n + ".conv2"] = pk.pack(d + ".conv2")
                if Lr.cin != Lr.cout:
                    w[n + ".nin_shortcut"] = pk.pack(d + ".nin_shortcut")
            elif Lr.kind == "attn":
                if Lr.ch not in (32, 64, 128, 256, 512):
                    raise NotImplementedError("VAE AttnBlock width %d" % Lr.ch)
                norm(n + ".norm")
                w[n + ".qkv"] = pk.pack([d + ".q", d + ".k", d + ".v"], n_out=2 * Lr.ch)
                w[n + ".proj_out"] = pk.pack(d + ".proj_out")
            elif Lr.kind == "upconv":
                w[n] = pk.add_upsample_phases(pk.pack(d), d)
            elif Lr.kind == "norm_out":
                norm(n)
        self.w, self.v = w, v


class VAEDecodePlan(Emitter):
    """decode_first_stage (ddpm.py:771-829 plain branch): z / scale_factor ->
    post_quant_conv -> Decoder (model.py:535-568) -> fp32 NCHW image."""

    def __init__(self, ctx, packed: PackedVAEDecoder, B, h, w, scale_factor):
        super().__init__(ctx)
        self.pk, self.arch = packed, packed.arch
        a = self.arch
        self.B, self.h, self.w = B, h, w
        f = a.factor
        self.z = self.alloc(B, a.embed_dim, h, w, dtype=torch.float32)
        self.img = self.alloc(B, a.out_ch, h * f, w * f, dtype=torch.float32)
        self.gn_ws = self.alloc(max(64, ctx.groupnorm_ws_bytes(B, h * f * w * f) // 4), dtype=torch.float32)
        self.prog = Program(ctx)
        P, W_, V_ = self.prog, packed.w, packed.v
        zin = Act(self.alloc(B * h * w, _rup(a.embed_dim, 32), zero=True), B, h, w, a.embed_dim)
        lib, hh, chk, z = self.lib, self.hctx, self._chk, self.z
        inv = 1.0 / float(scale_factor)
        P.add(lambda s: chk(lib.upk_nchw_f32_to_nhwc_f16(hh, z.data_ptr(), B, a.embed_dim, h * w, zin.t.data_ptr(),
                                                         zin.ld, 0, 0, inv, s)))
        x = self.conv(P, zin, W_["post_quant_conv"],
                      out=Act(self.alloc(B * h * w, _rup(a.z_channels, 32), zero=True), B, h, w, a.z_channels))
        for Lr in a.decoder:
            n = Lr.name
            if Lr.kind == "conv":
                x = self.conv(P, x, W_[n], gn_stats=VAE_GN_BYPRODUCT)
            elif Lr.kind == "resnet":
                h1 = self.conv(P, x, W_[n + ".conv1"], gn=(*V_[n + ".norm1"], 1e-6, True, self.gn_ws), gn_stats=VAE_GN_BYPRODUCT)
                sk = self.conv(P, x, W_[n + ".nin_shortcut"]) if Lr.cin != Lr.cout else x
                x = self.conv(P, h1, W_[n + ".conv2"], residual=sk, gn=(*V_[n + ".norm2"], 1e-6, True, self.gn_ws),
                              gn_stats=VAE_GN_BYPRODUCT)
            elif Lr.kind == "attn":
                c, HW = Lr.ch, x.H * x.W
                xn = self.groupnorm(P, x, *V_[n + ".norm"], 1e-6, False, self.gn_ws)
                qk = Act(self.alloc(x.M, 2 * c), x.B, x.H, x.W, 2 * c)
                vt_ld = _rup(HW, 32)
                vt = self.alloc(x.B, 1, c, vt_ld, zero=True)
                self.conv(P, xn, W_[n + ".qkv"], out=qk,
                          vt=dict(t=vt, heads=1, dhead=c, ld=vt_ld, tokens=HW, **{"from": 2 * c}))
                ao = Act(self.alloc(x.M, c), x.B, x.H, x.W, c)
                self.attention(P, qk.t, 2 * c, HW * 2 * c, qk.t[:, c:], 2 * c, HW * 2 * c, vt, vt_ld, ao.t, c, HW * c,
                               x.B, 1, HW, HW, c, int(c) ** -0.5)
                x = self.conv(P, ao, W_[n + ".proj_out"], residual=x, gn_stats=VAE_GN_BYPRODUCT)
            elif Lr.kind == "upconv":
                x = self.conv(P, x, W_[n], flags=L.F_UPSAMPLE2X)
            elif Lr.kind == "norm_out":
                x = self.groupnorm(P, x, *V_[n], 1e-6, True, self.gn_ws)
            elif Lr.kind == "conv_out":
                self.conv(P, x, W_[n], nchw_out=self.img)
        self.graph = None

    def run(self, z):
        z = z.to(self.dev, torch.float32).contiguous()
        require(tuple(z.shape) == tuple(self.z.shape), lambda: repr((z.shape, self.z.shape)), ValueError)
        self.z.copy_(z)
        self.prog.run()
        return self.img


# ====================================================================== VAE encoder
class PackedVAEEncoder:
    def __init__(self, ctx, arch: VAEArch, get):
        pk = Packer(ctx, get)
        self.arch = arch
        w, v = {}, {}

        def norm(name):
            v[name] = (pk.vec("encoder." + name + ".weight"), pk.vec("encoder." + name + ".bias"))

        for Lr in arch.encoder:
            n, d = Lr.name, "encoder." + Lr.name
            if Lr.kind in ("conv", "conv_out"):
                w[n] = pk.pack(d, cin_packed=_rup(Lr.cin, 32))
            elif Lr.kind == "resnet":
                norm(n + ".norm1")
                w[n + ".conv1"] = pk.pack(d + ".conv1")
                norm(n + ".norm2")
                w[n + ".conv2"] = pk.pack(d + ".conv2")
                if Lr.cin != Lr.cout:
                    w[n + ".nin_shortcut"] = pk.pack(d + ".nin_shortcut")
            elif Lr.kind == "attn":
                if Lr.ch not in (32, 64, 128, 256, 512):
                    raise NotImplementedError("VAE AttnBlock width %d" % Lr.ch)
                norm(n + ".norm")
                w[n + ".qkv"] = pk.pack([d + ".q", d + ".k", d + ".v"], n_out=2 * Lr.ch)
                w[n + ".proj_out"] = pk.pack(d + ".proj_out")
            elif Lr.kind == "downconv":
                w[n] = pk.pack(d)
            elif Lr.kind == "norm_out":
                norm(n)
        zc2 = arch.encoder[-1].cout
        w["quant_conv"] = pk.pack("quant_conv", cin_packed=_rup(zc2, 32))
        self.w, self.v = w, v


class VAEEncodePlan(Emitter):
    """AutoencoderKL.encode (autoencoder.py:324-328): Encoder (model.py:434-459) -> quant_conv ->
    posterior moments [B, 2*embed_dim, h, w] fp32 NCHW.  Same kernels as the decoder; the
    stride-2 Downsample uses the asymmetric (0,1,0,1) padding mode of the implicit GEMM."""

    def __init__(self, ctx, packed: PackedVAEEncoder, B, H, W):
        super().__init__(ctx)
        self.pk, self.arch = packed, packed.arch
        a = self.arch
        f = a.factor
        require(H % f == 0 and W % f == 0, lambda: "image size must be a multiple of %d" % f, ValueError)
        self.x = self.alloc(B, a.in_channels, H, W, dtype=torch.float32)
        self.moments = self.alloc(B, 2 * a.embed_dim, H // f, W // f, dtype=torch.float32)
        self.gn_ws = self.alloc(max(64, ctx.groupnorm_ws_bytes(B, H * W) // 4), dtype=torch.float32)
        self.prog = Program(ctx)
        P, W_, V_ = self.prog, packed.w, packed.v
        xin = Act(self.alloc(B * H * W, _rup(a.in_channels, 32), zero=True), B, H, W, a.in_channels)
        lib, hh, chk, xs = self.lib, self.hctx, self._chk, self.x
        P.add(lambda s: chk(lib.upk_nchw_f32_to_nhwc_f16(hh, xs.data_ptr(), B, a.in_channels, H * W,
                                                         xin.t.data_ptr(), xin.ld, 0, 0, 1.0, s)))
        x = xin
        for Lr in a.encoder:
            n = Lr.name
            if Lr.kind == "conv":
                x = self.conv(P, x, W_[n], gn_stats=VAE_GN_BYPRODUCT)
            elif Lr.kind == "resnet":
                h1 = self.conv(P, x, W_[n + ".conv1"], gn=(*V_[n + ".norm1"], 1e-6, True, self.gn_ws), gn_stats=VAE_GN_BYPRODUCT)
                sk = self.conv(P, x, W_[n + ".nin_shortcut"]) if Lr.cin != Lr.cout else x
                x = self.conv(P, h1, W_[n + ".conv2"], residual=sk, gn=(*V_[n + ".norm2"], 1e-6, True, self.gn_ws),
                              gn_stats=VAE_GN_BYPRODUCT)
            elif Lr.kind == "attn":
                c, HW = Lr.ch, x.H * x.W
                xn = self.groupnorm(P, x, *V_[n + ".norm"], 1e-6, False, self.gn_ws)
                qk = Act(self.alloc(x.M, 2 * c), x.B, x.H, x.W, 2 * c)
                vt_ld = _rup(HW, 32)
                vt = self.alloc(x.B, 1, c, vt_ld, zero=True)
                self.conv(P, xn, W_[n + ".qkv"], out=qk,
                          vt=dict(t=vt, heads=1, dhead=c, ld=vt_ld, tokens=HW, **{"from": 2 * c}))
                ao = Act(self.alloc(x.M, c), x.B, x.H, x.W, c)
                self.attention(P, qk.t, 2 * c, HW * 2 * c, qk.t[:, c:], 2 * c, HW * 2 * c, vt, vt_ld, ao.t, c, HW * c,
                               x.B, 1, HW, HW, c, int(c) ** -0.5)
                x = self.conv(P, ao, W_[n + ".proj_out"], residual=x)
            elif Lr.kind == "downconv":
                x = self.conv(P, x, W_[n], stride=2, flags=L.F_PAD_ASYM)
            elif Lr.kind == "norm_out":
                x = self.groupnorm(P, x, *V_[n], 1e-6, True, self.gn_ws)
            elif Lr.kind == "conv_out":
                x = self.conv(P, x, W_[n], out=Act(self.alloc(x.M, _rup(Lr.cout, 32), zero=True), x.B, x.H, x.W,
                                                   Lr.cout))
        self.conv(P, x, W_["quant_conv"], nchw_out=self.moments)

    def run(self, img):
        img = img.to(self.dev, torch.float32).contiguous()
        require(tuple(img.shape) == tuple(self.x.shape), lambda: repr((img.shape, self.x.shape)), ValueError)
        self.x.copy_(img)
        self.prog.run()
        return self.moments


# ====================================================================== sampler step graph
class SamplerState:
    """Device state of one DDIM run on a sampler-mode UNetPlan: latent x (fp32 NCHW),
    pred_x0, the per-step coefficient / noise tables and the captured step graph
    (UNet body -> upk_ddim_step_f32 -> upk_advance_step)."""

    def __init__(self, plan: UNetPlan, channels, cfg=False, plms=False):
        """cfg: classifier-free guidance — the plan runs 2*B rows ([unconditional ; conditional], ddim.py:173-178),
        the latent state has B = plan.B // 2 samples and the update combines the two halves of eps.
        plms: the graph is one PLMS model evaluation (upk_plms_step_f32; plan rows = evaluations = steps + 1)."""
        require(plan.mode == "sampler", "SamplerState needs a sampler-mode plan", ValueError)
        self.plan = plan
        self.cfg = bool(cfg)
        self.plms = bool(plms)
        require(not cfg or plan.B % 2 == 0, "guidance runs [uncond ; cond]: the plan's batch must be even", ValueError)
        B, H, W, R = (plan.B // 2 if cfg else plan.B), plan.H, plan.W, plan.rows
        self.B = B
        self.C = channels
        self.x = plan.alloc(B, channels, H, W, dtype=torch.float32)
        self.pred_x0 = plan.alloc(B, channels, H, W, dtype=torch.float32)
        self.coefs = plan.alloc(R, 4, dtype=torch.float32)
        self.noise = None
        self.graphs = {}
        self.step_done = plan.alloc(1, dtype=torch.int32, zero=True)  # arrival counter of the step kernels
        self.hist = plan.alloc(3, B * channels * H * W, dtype=torch.float32) if plms else None

    def close(self):
        """Destroys the instantiated HIP graphs (they hold device memory; called when the owning plan is dropped).
        sample() hands back a clone enqueued behind the last replay, so a replay may still be in flight when a new shape
        evicts this plan: the device is drained first (HIP does not promise deferred destruction of an executing graph)."""
        if self.graphs:
            torch.cuda.synchronize(self.plan.dev)
        for g in self.graphs.values():
            self.plan.ctx.graph_destroy(g)
        self.graphs = {}

    def ensure_noise(self):
        if self.noise is None:
            p = self.plan
            self.noise = p.alloc(p.rows, self.B * self.C * p.H * p.W, dtype=torch.float32)
        return self.noise

    def _emit_tail(self, stream, with_noise, scale=1.0):
        p = self.plan
        nz = self.noise.data_ptr() if with_noise else None
        # the step kernel also advances the device-side step counter (include/upk.h upk_step_autoadvance)
        p.ctx._chk(p.lib.upk_step_autoadvance(p.hctx, self.step_done.data_ptr()))
        try:
            self._emit_step(stream, nz, scale)
        finally:
            p.ctx._chk(p.lib.upk_step_autoadvance(p.hctx, None))

    def _emit_step(self, stream, nz, scale):
        p = self.plan
        if self.plms:
            require(nz is None, "PLMS runs with eta = 0", ValueError)
            p.ctx._chk(p.lib.upk_plms_step_f32(p.hctx, self.x.data_ptr(), p.eps.data_ptr(), self.coefs.data_ptr(),
                                               p.step.data_ptr(), self.hist.data_ptr(), self.pred_x0.data_ptr(),
                                               p.xin.t.data_ptr(), p.xin.ld, self.B, self.C, p.H * p.W, float(scale),
                                               int(self.cfg), stream))
        elif self.cfg:
            p.ctx._chk(p.lib.upk_ddim_step_cfg_f32(p.hctx, self.x.data_ptr(), p.eps.data_ptr(), self.coefs.data_ptr(), nz,
                                                   p.step.data_ptr(), self.pred_x0.data_ptr(), p.xin.t.data_ptr(),
                                                   p.xin.ld, self.B, self.C, p.H * p.W, float(scale), stream))
        else:
            p.ctx._chk(p.lib.upk_ddim_step_f32(p.hctx, self.x.data_ptr(), p.eps.data_ptr(), self.coefs.data_ptr(), nz,
                                               p.step.data_ptr(), self.pred_x0.data_ptr(), p.xin.t.data_ptr(), p.xin.ld,
                                               p.B, self.C, p.H * p.W, stream))

    def step_eager(self, with_noise, scale=1.0):
        s = self.plan.ctx._s()
        self.plan.body.run(s)
        self._emit_tail(s, with_noise, scale)

    def graph(self, with_noise, scale=1.0, nsteps=1):
        """`nsteps` consecutive DDIM steps captured as ONE HIP graph (static shapes, device-side step index: the same
        graph serves every position of the loop; the guidance scale is a kernel argument, so each scale value gets its
        own graph).  nsteps > 1 saves the graph-to-graph launch gap of the steps inside."""
        key = (with_noise, float(scale) if self.cfg else 1.0) + ((int(nsteps),) if nsteps != 1 else ())
        g = self.graphs.pop(key, None)
        if g is not None:
            self.graphs[key] = g  # (most recently used last: eviction takes the least recently used graph)
        if g is None:
            p = self.plan
            if with_noise:
                self.ensure_noise()
            side = torch.cuda.Stream(device=p.dev)
            side.wait_stream(torch.cuda.current_stream(p.dev))
            sp = side.cuda_stream
            p.ctx._chk(p.lib.upk_graph_begin(p.hctx, sp))
            try:
                for _ in range(int(nsteps)):
                    p.body.run(sp)
                    self._emit_tail(sp, with_noise, scale)
            finally:
                gh = C.c_void_p()
                rc = p.lib.upk_graph_end(p.hctx, sp, C.byref(gh))
            p.ctx._chk(rc)
            torch.cuda.current_stream(p.dev).wait_stream(side)
            # one sample() makes up to three graphs per (noise, scale) (full groups of steps, the remainder, single steps
            # around a callback): 16 graphs = five guidance scales in rotation; least recently used first
            if len(self.graphs) >= 16:
                torch.cuda.synchronize(p.dev)  # (the evicted graph may still be executing)
                p.ctx.graph_destroy(self.graphs.pop(next(iter(self.graphs))))
            g = self.graphs[key] = gh
        return g

    def launch(self, with_noise, scale=1.0, nsteps=1):
        p = self.plan
        p.ctx._chk(p.lib.upk_graph_launch(p.hctx, self.graph(with_noise, scale, nsteps), p.ctx._s()))

// Shared by igemm.hip, astat.hip, mlp.hip and xblock.hip: launch arguments, LDS swizzle and the epilogues of the implicit-GEMM kernels.
#pragma once
#include <stdlib.h>

#include "common.h"

namespace upkd {

// debug-only ablation bits (env UPK_ABLATE, read per launch): which phase owns the time?
enum {
  ABL_NOEPI = 0x10000,
  ABL_NOGLOAD = 0x20000,
  ABL_NOLDSW = 0x40000,
  ABL_NOMFMA = 0x80000,
  ABL_EMPTY = 0x100000,  // WS kernel returns at entry (pure launch cost of its geometry)
  ABL_TIMELINE = 0x200000,  // WS kernel: blocks 0 and gridDim.x-1 write s_memtime stamps to the workspace
  ABL_NOSTORE = 0x400000,   // plain epilogue: everything but the global stores of the output
  ABL_NOGNP = 0x800000,     // plain epilogue: no GroupNorm channel partials (column sums, LDS exchange, their stores)
  ABL_NOEPILD = 0x1000000,  // plain epilogue: operands (bias, row vector, residual) read as zeros, no loads
  ABL_NOSLAB = 0x2000000,   // split-K launches: the partial slabs (fp16, fp32 with -DUPK_SLAB_F32) are not written
  ABL_NOA = 0x4000000,      // WS loaders: every A (im2col) row group fetches the zero page — the requests stay, their lines do not
  ABL_NOB = 0x8000000,      // WS loaders: every B (weight) row group fetches the zero page
};
// The ablation / timeline hooks are compiled in only for dev builds (UPK_CXXFLAGS=-DUPK_DEV, scripts/ablate.sh,
// scripts/timeline*.py): even as never-taken runtime tests they cost scalar registers and instructions in the loops.
#if defined(UPK_DEV) || defined(UPK_TIMELINE)
#define ABL_ON(f) ((a.flags & (f)) != 0)
#else
#define ABL_ON(f) (false)
#endif


struct IgemmArgs {
  const f16* x1;
  const f16* x2;
  int c1, c2, ld1, ld2;
  const f16* w;
  const f16* zero;  // zero page (device)
  int npad;
  const float* bias;
  const f16* res;
  int ldr;
  const float* rowvec;
  int rv_bs, rv_ss;
  const int* step;
  void* y;
  int ldy;
  f16* vt;
  int vt_from, vt_heads, vt_dhead, vt_ld, vt_tokens;
  float* partial;  // split-K slabs [splitk][M][npad] of slab_t (fp16 unless -DUPK_SLAB_F32; typed float* for the ABI's workspace), or nullptr
  int M, n_out;
  int B, HS, WS;   // stored input dims
  int HL, WL;      // logical input dims (after optional 2x upsample)
  int Ho, Wo;
  int ks, stride, pad_lo, ups;
  int linear;      // ks == 1 && stride == 1 && !ups: rows are addressed directly
  unsigned long long* dbg;  // ABL_TIMELINE stamps
  const float* ln_u;  // folded LayerNorm: column sums of the packed (gamma-scaled) weight, or nullptr
  float* gn_cp;       // plain epilogue also writes per-(row block, channel) GroupNorm partials here, or nullptr
  int gn_nblk, gn_hw;  // row blocks (= M tiles) per sample, pixels per sample
  float ln_eps;
  float ln_inv_dim;
  int cpt;         // 32-wide chunks per tap = (c1+c2)/32
  int nchunks;     // ks*ks*cpt + appended chunks
  int nchunks_main;  // ks*ks*cpt: the appended 1x1 segment (x3 | x4 at the output pixel) starts here
  const f16* x3;
  const f16* x4;
  int c3, c4, ld3, ld4;
  int chunks_per_split;
  int tiles_m, tiles_n;
  int flags;
  int tile_rows;   // valid output rows per M tile (<= BM); 0: BM
  // ---- LayerNorm row statistics handed from the producer of a tensor to the folded-LayerNorm GEMM that reads it
  float* lnr_out;        // plain epilogue also writes per-(column slot, row) {sum, sum of squares} of what it stores:
                         // [slots][M][2], slot = (output column) / (columns per wave); nullptr: no
  const float* lnr_in;   // folded LayerNorm (ln_u) takes the row statistics from here ([lnr_slots][M][2]) instead of
  int lnr_slots;         // from the A fragments: any kernel family can run the GEMM
  // ---- conv3x3 behind a nearest 2x upsample as four 2x2 convs on the low-resolution grid (include/upk.h w_phase):
  // grid.y = phase (py, px) = (y >> 1, y & 1); phase weights at w + phase * ph_wstride, padding (1 - py, 1 - px),
  // row m = (b, y, x) of the low-resolution grid is written to output pixel (2y + py, 2x + px)
  int ph_on;
  int ph_wstride;
  // log2 of Ho*Wo / Wo when they are powers of two (the 256x256 workloads: 32, 16, 8, 4), else -1: the im2col row
  // decode and the epilogue's sample index become shifts instead of two ~20-instruction integer divisions per row,
  // on the path to the first DMA of every 3x3 launch
  int sh_hw, sh_w;
  // ---- XCD-aware tile order (xm_pm > 0): workgroup w runs on XCD w % 8; the 8 XCDs form a pm x pn grid over
  // (M tiles) x (N tile, K split) units, so that an operand is fetched over the fabric by pm (weights) / pn
  // (activations) XCDs instead of by every XCD that happens to hold one of its tiles.  mi x nj = tiles x units per
  // XCD; grid.x = 8 * mi * nj, surplus workgroups exit
  int xm_pm, xm_pn, xm_mi, xm_nj, xm_z;
  // ---- A-stationary family (astat.hip): a workgroup keeps its BM x K activation tile in LDS and walks passes
  // [tn * as_ppw, min(as_npass, (tn + 1) * as_ppw)) of 8 waves x NI x 16 output columns; tiles_n = N super tiles
  int as_ppw, as_npass;
  // ---- next-weight prefetch (include/upk.h pf_next): lines of pf[0 .. pf_lines * 128) are touched by this launch's workgroups
  const char* pf;
  int pf_lines;
  int pf_self;  // (dev experiment, UPK_SELF_PREFETCH=1: the workgroups of an XCD that share a weight slice touch it cooperatively up front)
};

// Split-K partial slabs ([split][M][n_pad] in the caller's workspace, IgemmArgs::partial).  fp16: the partial sums are
// rounded once on their way out and added in fp32 by the reduce pass, in slab order (deterministic as before) — half
// the bytes of the fp32 slabs, which cost 80 us of the forward on their way out alone (phase ablation,
// profiles/r04_forward_phase_ablation_igemm_ws.txt).  Error: one extra fp16 rounding per partial, ~|y| 2^-11 in
// total for z partials of magnitude |y| / sqrt z — the size of the output's own rounding.  -DUPK_SLAB_F32: fp32 slabs.
#ifdef UPK_SLAB_F32
typedef float slab_t;
typedef f32x4 slab4;
#else
typedef f16 slab_t;
typedef f16x4 slab4;
#endif
// write-through (sc1): the partials are on their way to memory when the kernel ends instead of being written back
// at the boundary (MI355X_MICROARCH.md publish-large / boundary); same-box A/B on the forward: 2.946 -> 2.941 ms
__device__ __forceinline__ void slab_store(slab_t* p, f32x4 v) {
#ifdef UPK_SLAB_F32
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#else
  // saturating: a partial beyond the fp16 range (K slices of large, cancelling activations) stays finite, so the fp32 sum
  // of the slices is off by the clipped amount instead of inf / nan (v_med3_f32: one VALU instruction per element)
  const f16x4 h = {(f16)__builtin_amdgcn_fmed3f(v[0], -65504.f, 65504.f), (f16)__builtin_amdgcn_fmed3f(v[1], -65504.f, 65504.f),
                   (f16)__builtin_amdgcn_fmed3f(v[2], -65504.f, 65504.f), (f16)__builtin_amdgcn_fmed3f(v[3], -65504.f, 65504.f)};
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(h) : "memory");
#endif
}
__device__ __forceinline__ f32x4 slab_load4(const slab_t* p) {
  const slab4 v = *(const slab4*)p;
  return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

// (tm, tn, split index) of this workgroup; false: nothing to do (XCD-aware grids are padded)
__device__ __forceinline__ bool tile_map(const IgemmArgs& a, int& tm, int& tn, int& zs) {
  if (a.xm_pm == 0) {
    const int tile = blockIdx.x;
    tn = tile / a.tiles_m;
    tm = tile - tn * a.tiles_m;
    zs = blockIdx.z;
    return true;
  }
  const int w = blockIdx.x;
  const int xcd = w & 7, l = w >> 3;
  const int xi = xcd / a.xm_pn, xj = xcd - xi * a.xm_pn;
  const int ul = l / a.xm_mi, tl = l - ul * a.xm_mi;  // M tiles fastest: neighbours in time share the weight slice
  tm = xi * a.xm_mi + tl;
  const int u = xj * a.xm_nj + ul;
  zs = u / a.tiles_n;
  tn = u - zs * a.tiles_n;
  return tm < a.tiles_m && zs < a.xm_z;
}

__device__ __forceinline__ int div_hw(const IgemmArgs& a, int m, int hw) { return a.sh_hw >= 0 ? m >> a.sh_hw : m / hw; }
__device__ __forceinline__ int div_w(const IgemmArgs& a, int p) { return a.sh_w >= 0 ? p >> a.sh_w : p / a.Wo; }

// phase-mode helpers (uniform: blockIdx.y)
__device__ __forceinline__ int ph_id(const IgemmArgs& a) { return a.ph_on ? (int)blockIdx.y : 0; }
__device__ __forceinline__ unsigned ph_row(const IgemmArgs& a, unsigned mm) {  // output row of low-resolution row mm
  const unsigned W = (unsigned)a.Wo;
  const unsigned q = a.sh_w >= 0 ? mm >> a.sh_w : mm / W, x = mm - q * W, ph = blockIdx.y;
  return (2u * q + (ph >> 1)) * 2u * W + 2u * x + (ph & 1u);
}

// 16-byte chunk swizzle for a [rows][4 chunks] fp16 tile (64 B rows).
// ds_read_b128 is serviced in four 16-lane groups {0-3,12-15,20-27},{4-11,16-19,28-31},...
// With fragment lane l reading row (l&15), chunk (l>>4), XOR-ing the chunk with
// (-(row>>2))&3 puts the 16 lanes of every group on 16 distinct 16-B slots of the
// 256-B bank row.
__device__ __forceinline__ int lds_swz(int row, int chunk) { return chunk ^ ((-(row >> 2)) & 3); }

// Epilogue, split in a per-ROW part (integer division for the sample index, row offsets:
// once per 16-row fragment) and a per-4-COLUMN part, with 32-bit offsets against uniform
// base pointers — with K loops as short as 7..16 chunks the epilogue is a large share of
// the issued instructions, so it is kept lean.
struct RowCtx {
  bool ok;
  unsigned y_off, res_off, rv_off, vt_off, nchw_off;
};

// sum over the four lanes (c, c+16, c+32, c+48) that hold the columns of one row of a fragment (gfx950 permlane swaps:
// VALU, no LDS round trip); every lane ends with the total
__device__ __forceinline__ float sum_lane_groups(float x) {
  unsigned u = __builtin_bit_cast(unsigned, x);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]));
  const auto b = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}

struct Epi {
  // folded LayerNorm with the row statistics from memory (IgemmArgs::lnr_in): {mean, rstd} of row m
  static __device__ __forceinline__ void lnr_row(const IgemmArgs& a, int m, float& mean, float& rstd) {
    const float* p = a.lnr_in + (long)(m < a.M ? m : 0) * 2;
    const long sstride = (long)a.M * 2;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {  // (all loads issued before the first use)
      const f32x2 v = s < a.lnr_slots ? *(const f32x2*)(p + s * sstride) : (f32x2){0.f, 0.f};
      s1 += v[0];
      s2 += v[1];
    }
    mean = s1 * a.ln_inv_dim;
    rstd = rsqrtf(fmaxf(s2 * a.ln_inv_dim - mean * mean, 0.f) + a.ln_eps);
  }
  // y = rstd * (acc - mean * colsum) on a wave's accumulator tile (lane (lc, lg): row mw + 16 i + lc).  The loads sit
  // in the epilogue: holding {mean, rstd} of the MI row groups across the K loop instead (requested at kernel start)
  // costs every launch of the kernel 2*MI registers — measured: the 3x3 224->224 conv with residual 22 -> 31 us
  template <int MI, int NI>
  static __device__ __forceinline__ void lnr_fix(const IgemmArgs& a, int mw, int nw, int lc, int lg, f32x4 (&acc)[MI][NI]) {
    f32x4 u[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = nw + j * 16 + lg * 4;
      u[j] = n < a.npad ? *(const f32x4*)(a.ln_u + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float mean[MI], rstd[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) lnr_row(a, mw + i * 16 + lc, mean[i], rstd[i]);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = (acc[i][j] - mean[i] * u[j]) * rstd[i];
  }
  static __device__ __forceinline__ f32x4 lnr_fix1(const IgemmArgs& a, int m, int n, f32x4 v) {
    float mean, rstd;
    lnr_row(a, m, mean, rstd);
    const f32x4 u = n < a.npad ? *(const f32x4*)(a.ln_u + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    return (v - mean * u) * rstd;
  }

  static __device__ __forceinline__ RowCtx row(const IgemmArgs& a, int m, int mlim) {
    RowCtx r;
    r.ok = m < mlim;
    const int mm = r.ok ? m : 0;
    r.y_off = (a.ph_on ? ph_row(a, (unsigned)mm) : (unsigned)mm) * (unsigned)a.ldy;
    r.res_off = (unsigned)mm * (unsigned)a.ldr;
    r.rv_off = 0;
    r.vt_off = 0;
    r.nchw_off = 0;
    if (a.rowvec || (a.flags & UPK_F_OUT_NCHW_F32)) {
      const int hw = a.Ho * a.Wo;
      const int b = div_hw(a, mm, hw);
      const int p = mm - b * hw;
      const int st = (a.rowvec && a.step) ? *a.step : 0;
      r.rv_off = (unsigned)(st * a.rv_ss + b * a.rv_bs);
      r.nchw_off = (unsigned)(b * a.n_out * hw + p);
    }
    if (a.vt) {
      const int bb = mm / a.vt_tokens;
      const int tok = mm - bb * a.vt_tokens;
      r.vt_off = (unsigned)(bb * a.vt_heads * a.vt_dhead * a.vt_ld + tok);
    }
    return r;
  }

  // Epilogue operands of one (row, 4-column) fragment.  They are FETCHED for a whole batch of
  // fragments before the first store of the batch: with load -> convert -> store per fragment the
  // stores (which may alias the residual as far as the compiler knows) serialise the loads, and a
  // 7-fragment epilogue costs seven dependent L2/fabric round trips (measured 2.2-3.5 us of a
  // 5-16 us launch with in-kernel s_memtime stamps).
  struct In {
    f32x4 rv;
    f16x4 res;
  };
  static __device__ __forceinline__ f32x4 bias4(const IgemmArgs& a, int n) {
    return (a.bias && n < a.npad) ? *(const f32x4*)(a.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  static __device__ __forceinline__ int out_col(const IgemmArgs& a, int n) {
    return (a.flags & UPK_F_GEGLU) ? (n >> 6) * 32 + (n & 31) : n;
  }
  static __device__ __forceinline__ In fetch(const IgemmArgs& a, const RowCtx& r, int n) {
    In in;
    in.rv = (f32x4){0.f, 0.f, 0.f, 0.f};
    in.res = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    if (!r.ok || n >= a.npad) return in;
    const int oc = out_col(a, n);
    const bool to_vt = a.vt && n >= a.vt_from;
    if (oc >= a.n_out && !to_vt) return in;
    if (a.rowvec) in.rv = *(const f32x4*)(a.rowvec + r.rv_off + n);
    if (a.res && !to_vt) in.res = *(const f16x4*)(a.res + r.res_off + oc);
    return in;
  }

  // finishes and stores packed columns [n, n+4) of the row; v = value accumulators, g = gate
  // (GEGLU) with their biases bv / bg, `in` = the operands fetched above
  static __device__ __forceinline__ void store(const IgemmArgs& a, const RowCtx& r, int n, f32x4 v, f32x4 g,
                                               const f32x4 bv, const f32x4 bg, const In& in) {
    if (!r.ok) return;
    const int flags = a.flags;
    v += bv;
    if (flags & UPK_F_GEGLU) {
      // packed rows: [32 value | 32 gate] per 64-row block
      g += bg;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = upk_geglu_mul(v[k], g[k]);
    }
    const int oc = out_col(a, n);  // output column
    const bool to_vt = a.vt && n >= a.vt_from;
    if (oc >= a.n_out && !to_vt) return;
    if (a.rowvec) v += in.rv;
    if (flags & UPK_F_SILU) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = upk_silu(v[k]);
    }
    if (flags & UPK_F_QUICKGELU) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v[k]));
    }
    if (to_vt) {
      const int cc = n - a.vt_from;  // = h * dhead + d  ->  row (h*dhead + d) of this sample's V^T
      f16* dst = a.vt + r.vt_off + (unsigned)cc * (unsigned)a.vt_ld;
#pragma unroll
      for (int k = 0; k < 4; ++k) dst[(unsigned)k * (unsigned)a.vt_ld] = (f16)v[k];
      return;
    }
    if (a.res) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += (float)in.res[k];
    }
    if (flags & UPK_F_OUT_NCHW_F32) {
      float* yo = (float*)a.y + r.nchw_off;
      const unsigned hw = (unsigned)(a.Ho * a.Wo);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (oc + k < a.n_out) yo[(unsigned)(oc + k) * hw] = v[k];
    } else if (flags & UPK_F_OUT_F32) {
      float* yo = (float*)a.y + r.y_off + oc;
      if (oc + 3 < a.n_out) {
        *(f32x4*)yo = v;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (oc + k < a.n_out) yo[k] = v[k];
      }
    } else {
      f16* yo = (f16*)a.y + r.y_off + oc;
      if (oc + 3 < a.n_out) {
        f16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (f16)v[k];
        *(f16x4*)yo = o;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (oc + k < a.n_out) yo[k] = (f16)v[k];
      }
    }
  }

  // The common epilogues as STRAIGHT-LINE code.  Every launch starts with a cold instruction cache,
  // and the general `store` above is a chain of taken branches over the GEGLU / SiLU / V^T / NCHW
  // blocks: ~8 jumps to cold lines per fragment, 5-8k cycles for a 4..7-fragment tile (s_memtime
  // stamps, scripts/timeline.py) against ~1k for the stores themselves.
  //   plain: fp32 acc + bias + timestep row vector + residual -> fp16 NHWC; absent operands point at
  //          the zero page instead of being branched around;
  //   geglu: (acc_v + b_v) * gelu(acc_g + b_g) -> fp16.
  static __host__ __device__ __forceinline__ bool plain(const IgemmArgs& a) {
    return !(a.flags & (UPK_F_GEGLU | UPK_F_SILU | UPK_F_QUICKGELU | UPK_F_OUT_F32 | UPK_F_OUT_NCHW_F32)) && !a.vt &&
           !(a.n_out & 3);
  }
  static __host__ __device__ __forceinline__ bool plain_geglu(const IgemmArgs& a) {
    return (a.flags & (UPK_F_GEGLU | UPK_F_SILU | UPK_F_QUICKGELU | UPK_F_OUT_F32 | UPK_F_OUT_NCHW_F32)) == UPK_F_GEGLU &&
           !a.vt &&
           !a.rowvec && !a.res && !(a.n_out & 3);
  }
  struct Plain {
    const float* bias;
    const float* rvp;
    const f16* resp;
    unsigned has_b, has_rv, has_res;
    int st, hw;
    __device__ __forceinline__ Plain(const IgemmArgs& a) : Plain(a, (a.rowvec && a.step) ? *a.step : 0) {}
    // (st_: the step index read by the caller, e.g. at kernel entry — the load is a dependent round trip here)
    __device__ __forceinline__ Plain(const IgemmArgs& a, int st_) {
      bias = a.bias ? a.bias : (const float*)a.zero;
      rvp = a.rowvec ? a.rowvec : (const float*)a.zero;
      resp = a.res ? a.res : a.zero;
      has_b = a.bias ? ~0u : 0u;
      has_rv = a.rowvec ? ~0u : 0u;
      has_res = a.res ? ~0u : 0u;
      st = st_;
      hw = a.Ho * a.Wo;
    }
    __device__ __forceinline__ f32x4 bias4(const IgemmArgs& a, int n) const {
      if ABL_ON(ABL_NOEPILD) return (f32x4){0.f, 0.f, 0.f, 0.f};
      return *(const f32x4*)(bias + ((n < a.npad ? (unsigned)n : 0u) & has_b));
    }
    // row part: offsets of row m (clamped to a valid row; the store is predicated on m < M)
    struct Row {
      bool ok;
      unsigned rv_off, res_off, y_off;
    };
    __device__ __forceinline__ Row row(const IgemmArgs& a, int m, int mlim) const {
      Row r;
      r.ok = m < mlim;
      const unsigned mm = r.ok ? (unsigned)m : 0u;
      r.rv_off = has_rv ? (unsigned)(st * a.rv_ss + div_hw(a, (int)mm, hw) * a.rv_bs) : 0u;
      r.res_off = (mm * (unsigned)a.ldr) & has_res;
      r.y_off = (a.ph_on ? ph_row(a, mm) : mm) * (unsigned)a.ldy;
      return r;
    }
    __device__ __forceinline__ f32x4 rv4(const IgemmArgs& a, const Row& r, int n) const {
      if ABL_ON(ABL_NOEPILD) return (f32x4){0.f, 0.f, 0.f, 0.f};
      return *(const f32x4*)(rvp + ((r.rv_off + (n < a.n_out ? (unsigned)n : 0u)) & has_rv));
    }
    __device__ __forceinline__ f16x4 res4(const IgemmArgs& a, const Row& r, int n) const {
      if ABL_ON(ABL_NOEPILD) return (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
      return *(const f16x4*)(resp + r.res_off + ((n < a.n_out ? (unsigned)n : 0u) & has_res));
    }
    static __device__ __forceinline__ f16x4 put(const IgemmArgs& a, const Row& r, int n, f32x4 v, const f16x4 rr) {
      f16x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (f16)(v[k] + (float)rr[k]);
      if (r.ok && n < a.n_out && !ABL_ON(ABL_NOSTORE)) *(f16x4*)((f16*)a.y + r.y_off + n) = o;
      return o;
    }
  };

  // sum over the 16 lanes of a DPP row (= the 16 rows a fragment's lanes with equal lg hold): rotate-add, every
  // lane ends with the total
  static __device__ __forceinline__ float row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
  }

  // tile_plain_cp: the tile also leaves the GroupNorm partial sums of what it stores: per (M tile, channel)
  // sum / sum of squares of the fp16-rounded outputs — in-lane over the MI row fragments, rotate-add over
  // the 16 rows of a fragment, through `red` (LDS, >= WM*WN*NI*32 floats) over the WM waves of a column, one
  // fixed order -> bitwise reproducible.  The host guarantees an M tile lies inside one sample (BM | H*W).
  template <int MI, int NI>
  static __device__ __forceinline__ void tile_plain(const IgemmArgs& a, int mw, int nw, int lc, int lg,
                                                    const f32x4 (&acc)[MI][NI], int mlim) {
    const Plain P(a);
    f32x4 bv[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) bv[j] = P.bias4(a, nw + j * 16 + lg * 4);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const Plain::Row r = P.row(a, mw + i * 16 + lc, mlim);
      f32x4 rv[NI];
      f16x4 rr[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        rv[j] = P.rv4(a, r, nw + j * 16 + lg * 4);
        rr[j] = P.res4(a, r, nw + j * 16 + lg * 4);
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) Plain::put(a, r, nw + j * 16 + lg * 4, acc[i][j] + bv[j] + rv[j], rr[j]);
    }
  }

  // tile_plain that also leaves the LayerNorm row sums of what it stores (IgemmArgs::lnr_out): per row the sum / sum
  // of squares of the fp16-rounded outputs over this wave's NI*16 columns — in-lane over the fragments, permlane swaps
  // over the four lane groups — one partial slot per wave column; the consumer adds the slots of a row
  template <int MI, int NI>
  static __device__ __forceinline__ void tile_plain_lnr(const IgemmArgs& a, int mw, int nw, int lc, int lg,
                                                        const f32x4 (&acc)[MI][NI], int mlim) {
    const Plain P(a);
    f32x4 bv[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) bv[j] = P.bias4(a, nw + j * 16 + lg * 4);
    float* dst = a.lnr_out + (long)(nw / (NI * 16)) * a.M * 2;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const Plain::Row r = P.row(a, mw + i * 16 + lc, mlim);
      f32x4 rv[NI];
      f16x4 rr[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        rv[j] = P.rv4(a, r, nw + j * 16 + lg * 4);
        rr[j] = P.res4(a, r, nw + j * 16 + lg * 4);
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = nw + j * 16 + lg * 4;
        const f16x4 o = Plain::put(a, r, n, acc[i][j] + bv[j] + rv[j], rr[j]);
        if (n < a.n_out) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float f = (float)o[k];
            s1 += f;
            s2 += f * f;
          }
        }
      }
      s1 = sum_lane_groups(s1);
      s2 = sum_lane_groups(s2);
      if (lg == 0 && r.ok) *(f32x2*)(dst + (long)(mw + i * 16 + lc) * 2) = (f32x2){s1, s2};
    }
  }

  // (separate from tile_plain: the extra accumulators and the workgroup barrier must not weigh on every launch)
  // NIT / jo: the NI fragments are columns [jo, jo + NI) of a wave column that is NIT fragments wide (a register tile
  // finished by two waves, igemm_ws_kernel's shared epilogue): `red` is indexed by the position in the whole column
  template <int MI, int NI, int WM, int WN, int NIT = NI>
  static __device__ __forceinline__ void tile_plain_cp(const IgemmArgs& a, int m0, int mw, int nw, int lc, int lg,
                                                       const f32x4 (&acc)[MI][NI], int wm, int wn, float* red,
                                                       int mlim, int jo = 0) {
    const Plain P(a);
    constexpr bool cp = true;
    f32x4 bv[NI];
    f32x4 cs[NI], cq[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      bv[j] = P.bias4(a, nw + j * 16 + lg * 4);
      cs[j] = cq[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const Plain::Row r = P.row(a, mw + i * 16 + lc, mlim);
      f32x4 rv[NI];
      f16x4 rr[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        rv[j] = P.rv4(a, r, nw + j * 16 + lg * 4);
        rr[j] = P.res4(a, r, nw + j * 16 + lg * 4);
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const f16x4 o = Plain::put(a, r, nw + j * 16 + lg * 4, acc[i][j] + bv[j] + rv[j], rr[j]);
        if (cp) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float f = r.ok ? (float)o[k] : 0.f;  // (rows past the tile's valid range)
            cs[j][k] += f;
            cq[j][k] += f * f;
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        cs[j][k] = row_sum16(cs[j][k]);
        cq[j][k] = row_sum16(cq[j][k]);
      }
    // red[(wm * WN + wn)][j][which][lg * 4 + k]
    float* mine = red + ((wm * WN + wn) * NIT + jo) * 32 + lg * 4;
    if (lc == 0) {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        *(f32x4*)(mine + j * 32) = cs[j];
        *(f32x4*)(mine + j * 32 + 16) = cq[j];
      }
    }
    __syncthreads();
    if (wm == 0 && lc == 0) {
      const int b = m0 / a.gn_hw;
      const int blk = (m0 - b * a.gn_hw) / (a.tile_rows > 0 ? a.tile_rows : MI * 16 * WM);
      float* dst = a.gn_cp + (long)((b * a.gn_nblk + blk) * 2) * a.npad;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = nw + j * 16 + lg * 4;
        if (n >= a.npad) continue;
        f32x4 su = {0.f, 0.f, 0.f, 0.f}, sq = su;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          su += *(const f32x4*)(red + ((w * WN + wn) * NIT + jo + j) * 32 + lg * 4);
          sq += *(const f32x4*)(red + ((w * WN + wn) * NIT + jo + j) * 32 + 16 + lg * 4);
        }
        *(f32x4*)(dst + n) = su;
        *(f32x4*)(dst + a.npad + n) = sq;
      }
    }
  }

  template <int MI, int NI>
  static __device__ __forceinline__ void tile_geglu(const IgemmArgs& a, int mw, int nw, int lc, int lg,
                                                    const f32x4 (&acc)[MI][NI], int mlim) {
    static_assert(NI % 4 == 0, "GEGLU tiles are [32 value | 32 gate] column blocks");
    const float* bias = a.bias ? a.bias : (const float*)a.zero;
    const unsigned has_b = a.bias ? ~0u : 0u;
    f32x4 bv[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = nw + j * 16 + lg * 4;
      bv[j] = *(const f32x4*)(bias + ((n < a.npad ? (unsigned)n : 0u) & has_b));
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = mw + i * 16 + lc;
      const bool ok = m < mlim;
      f16* yrow = (f16*)a.y + (ok ? (unsigned)m : 0u) * (unsigned)a.ldy;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        if (j & 2) continue;  // gate fragments are consumed by their value partner j - 2
        const int n = nw + j * 16 + lg * 4;
        const int oc = (n >> 6) * 32 + (n & 31);
        const f32x4 v = acc[i][j] + bv[j];
        const f32x4 g = acc[i][j + 2 < NI ? j + 2 : j] + bv[j + 2 < NI ? j + 2 : j];
        f16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (f16)upk_geglu_mul(v[k], g[k]);
        if (ok && n < a.npad && oc < a.n_out) *(f16x4*)(yrow + oc) = o;
      }
    }
  }

  // Epilogue of a wave's MI x NI register tile at (mw, nw).
  template <int MI, int NI, int WM, int WN>
  static __device__ __forceinline__ void tile(const IgemmArgs& a, int m0, int mw, int nw, int lc, int lg,
                                              const f32x4 (&acc)[MI][NI], int wm, int wn, float* red, int mlim) {
    if (plain(a)) {
      if (a.gn_cp && !ABL_ON(ABL_NOGNP)) tile_plain_cp<MI, NI, WM, WN>(a, m0, mw, nw, lc, lg, acc, wm, wn, red, mlim);
      else if (a.lnr_out) tile_plain_lnr<MI, NI>(a, mw, nw, lc, lg, acc, mlim);
      else tile_plain<MI, NI>(a, mw, nw, lc, lg, acc, mlim);
      return;
    }
    if constexpr (NI % 4 == 0) {
      if (plain_geglu(a)) {
        tile_geglu<MI, NI>(a, mw, nw, lc, lg, acc, mlim);
        return;
      }
    }
    const bool geglu = a.flags & UPK_F_GEGLU;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const RowCtx rc = row(a, mw + i * 16 + lc, mlim);
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = nw + j * 16 + lg * 4;
        if (n >= a.npad) continue;
        if (geglu) {
          if constexpr (NI % 4 == 0) {
            if ((j & 2) == 0) store(a, rc, n, acc[i][j], acc[i][j + 2 < NI ? j + 2 : j], bias4(a, n), bias4(a, n + 32),
                                    fetch(a, rc, n));
          }
        } else {
          const f32x4 b = bias4(a, n);
          store(a, rc, n, acc[i][j], acc[i][j], b, b, fetch(a, rc, n));
        }
      }
    }
  }
};


// astat.hip: the A-stationary family (configurations kNumCfgs .. of upk_conv_config_name)
struct AsPlan {
  int bm, pw, npass, ppw, tiles_m, tiles_n, lds_bytes;
};
int astat_num_configs();
const char* astat_config_name(int c);
int astat_config_ni(int c);
bool astat_plan(const upk_ctx* ctx, const IgemmArgs& a, int c, int ppw_req, AsPlan* pl);
int astat_launch(upk_ctx* ctx, IgemmArgs& a, int c, const AsPlan& pl, dim3 grid, hipStream_t stream);

// bigtile.hip: the big-tile family (configurations behind the A-stationary ones): 4 waves, one per SIMD, 256x256 /
// 256x128 / 128x256 block tiles for launches with at least one tile per CU
int bt_num_configs();
const char* bt_config_name(int c);
void bt_tile(int c, int* bm, int* bn, int* occ, int* mi, int* ni, int* wn);
bool bt_full_epilogue(int c);  // the configuration also exists with the general epilogue (Epi::tile)
int bt_launch(upk_ctx* ctx, const IgemmArgs& a, int c, dim3 grid, hipStream_t stream);



}  // namespace upkd

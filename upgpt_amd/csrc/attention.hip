// Fused attention for gfx950: softmax(Q K^T * scale) V with online softmax; the
// [B*h, n, n] score tensor of the reference (attention.py:180-191, model.py:185-195) never
// exists.  One wave owns 16 query rows; both matmuls run on v_mfma_f32_16x16x32_f16 with
// the operands arranged so that no cross-lane data movement is needed between them:
//
//   S^T[key][q]  = sum_d K[key][d] * Q[q][d]      A operand = K rows, B operand = Q rows
//        -> lane (g = lane>>4, c = lane&15) holds S^T[16t + 4g + r][c], t = 0,1, r = 0..3
//   O^T[dc][q]   = sum_k' V^T[dc][key(k')] * P^T[key(k')][q]
//        -> the B operand wants P^T[k' = 8g .. 8g+7][c]; we DEFINE key(8g + 4t + r) =
//           16t + 4g + r, i.e. exactly the 8 probabilities the lane already holds, and read
//           the V^T operand with the same permutation (two 8-byte loads per fragment).
//
// V arrives transposed ([B, heads, d, vt_ld]) straight from the projection GEMM's epilogue
// (igemm.hip, vt_* fields), K/Q are token-major with heads side by side.  K/V fragments are
// read through L1/L2 (per (b, head) they are <= 200 KB and shared by all query tiles).
#include <stdlib.h>

#include "common.h"

namespace {

struct AttnArgs {
  const f16* q;
  const f16* k;
  const f16* vt;
  f16* o;
  int ldq, ldk, vt_ld, ldo;
  long qbs, kbs, obs;
  int heads, nq, nkv;
  float scale_log2;  // scale * log2(e)
  int causal;        // key j attends to query i only if j <= i (CLIP text encoder); n_q == n_kv
};

// One K/V tile of NS*16 keys for the QT*16 queries of this wave (NS = 4 in the main loop, 2
// for the tail).  Every K / V^T fragment is loaded ONCE and used for all QT query groups (the
// level-1 self-attention is bound by K/V re-reads through L1/L2, not by MFMA).  Scores stay
// unscaled; softmax uses exp2(s*c - m*c) with c = scale*log2(e) folded into one FMA per
// score.  V^T fragments are requested before the softmax arithmetic so their latency hides
// under it.
// v_exp_f32 as it is: exp2f() wraps it in a range test, two selects and an ldexp so that results below 2^-126 come out
// as denormals; a softmax weight that small contributes nothing (arguments here are <= 0, -inf for masked keys -> 0),
// and the softmax is bound by VALU issue (d = 32: 9 of 13 instructions per score were that wrapper and the
// accumulator moves)
__device__ __forceinline__ float raw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// max over the four lanes (c, c+16, c+32, c+48) that hold the keys of one query: gfx950's permlane swaps are VALU
// instructions (2 + 2 issue slots), where __shfl_xor is two dependent LDS-crossbar round trips in the middle of the
// softmax's critical path
__device__ __forceinline__ float max_over_key_groups(float x) {
  unsigned u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u = __float_as_uint(fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])));
  const auto b = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
// sum over the four lanes (c, c+16, c+32, c+48) (see max_over_key_groups)
__device__ __forceinline__ float sum_over_key_groups(float x) {
  unsigned u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u = __float_as_uint(__uint_as_float(a[0]) + __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
#define ATT_ONES ((f16x8){(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f})

template <int D, int QREG, int NS, int QT>
__device__ __forceinline__ void attn_tile(const AttnArgs& a, const f16* const* qrow, const bool* q_ok,
                                          const f16x8 (*qf)[QREG ? D / 32 : 1], const f16* kbase,
                                          const f16* vbase, int kb, int g, int c, f32x4 (*o)[D / 16], float* mrun,
                                          f32x4* lsum, int q0) {
  constexpr int KD = D / 32;
  constexpr int DT = D / 16;
  constexpr int NC = NS / 2;  // 32-key chunks for the PV MFMAs
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  f32x4 s[QT][NS];
  const f16* kp[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    const int key = kb + 16 * t + c;
    kp[t] = kbase + (long)(key < a.nkv ? key : 0) * a.ldk;
#pragma unroll
    for (int u = 0; u < QT; ++u) s[u][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int kd = 0; kd < KD; ++kd) {
    f16x8 qv[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u)
      qv[u] = QREG ? qf[u][kd] : (q_ok[u] ? *(const f16x8*)(qrow[u] + kd * 32) : zero8);
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      const f16x8 kf = *(const f16x8*)(kp[t] + kd * 32);
#pragma unroll
      for (int u = 0; u < QT; ++u) s[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qv[u], s[u][t], 0, 0, 0);
    }
  }
  // V^T operand fragments (only when they fit in registers next to the accumulators)
  constexpr bool VPRE = (D <= 128);
  f16x4 vpre[VPRE ? DT * NC * 2 : 1];
  if (VPRE) {
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const f16* vr = vbase + (long)i * 16 * a.vt_ld + kb + 32 * j;
        vpre[(i * NC + j) * 2] = *(const f16x4*)vr;
        vpre[(i * NC + j) * 2 + 1] = *(const f16x4*)(vr + 16);
      }
  }
  const float cs = a.scale_log2;
  f16x8 pf[QT][NC];
  const bool tail = kb + NS * 16 > a.nkv;  // wave-uniform
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    // lane (g, c) holds keys kb + 16t + 4g + r of query c of group u
    if (tail || a.causal) {
      const int qlim = a.causal ? q0 + u * 16 + c : a.nkv;  // last key this query may see
#pragma unroll
      for (int t = 0; t < NS; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kb + 16 * t + 4 * g + r;
          if (key >= a.nkv || key > qlim) s[u][t][r] = -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NS; ++t)
      mx = fmaxf(mx, fmaxf(fmaxf(s[u][t][0], s[u][t][1]), fmaxf(s[u][t][2], s[u][t][3])));
    mx = max_over_key_groups(mx);
    const float mnew = fmaxf(mrun[u], mx);  // finite: every tile has >= 1 valid key
    // rescale only when some row's running maximum moved (wave-uniform test; alpha is exactly 1 otherwise): after the
    // first tiles it rarely does, and the accumulators then stay untouched in their MFMA registers
    if (__builtin_amdgcn_ballot_w64(mnew != mrun[u]) != 0) {
      const float alpha = raw_exp2((mrun[u] - mnew) * cs);
      lsum[u] *= alpha;
#pragma unroll
      for (int i = 0; i < DT; ++i) o[u][i] *= alpha;
    }
    mrun[u] = mnew;
    const float mc = -mnew * cs;
#pragma unroll
    for (int t = 0; t < NS; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) pf[u][t >> 1][(t & 1) * 4 + r] = (f16)raw_exp2(fmaf(s[u][t][r], cs, mc));
  }
  // softmax denominators on the MFMA pipe: an all-ones V^T fragment makes every row of the product the sum of the
  // (fp16) weights of the lane's query — no VALU adds, no cross-lane reduction, and numerator and denominator see the
  // same rounded weights
#pragma unroll
  for (int j = 0; j < NC; ++j)
#pragma unroll
    for (int u = 0; u < QT; ++u) lsum[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ATT_ONES, pf[u][j], lsum[u], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      f16x4 va, vb;
      if (VPRE) {
        va = vpre[(i * NC + j) * 2];
        vb = vpre[(i * NC + j) * 2 + 1];
      } else {
        const f16* vr = vbase + (long)i * 16 * a.vt_ld + kb + 32 * j;
        va = *(const f16x4*)vr;
        vb = *(const f16x4*)(vr + 16);
      }
      const f16x8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
#pragma unroll
      for (int u = 0; u < QT; ++u) o[u][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[u][j], o[u][i], 0, 0, 0);
    }
}

// grid.x = ceil(n_q / (64*QT)), grid.y = B*heads; 4 waves, each QT groups of 16 query rows.
template <int D, int QREG, int QT>
__global__ __launch_bounds__(256) void attn_kernel(const AttnArgs a) {
  constexpr int KD = D / 32;  // k-steps of Q K^T
  constexpr int DT = D / 16;  // output sub-tiles
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int bh = blockIdx.x;  // (all query blocks of one (sample, head) on one XCD: its K / V cross the fabric once)
#ifndef UPK_NO_XCD_SAMPLE
  const int nb_ = gridDim.x / a.heads;
  const int b = bh % nb_;
  const int h = bh / nb_;
#else
  const int b = bh / a.heads;
  const int h = bh - b * a.heads;
#endif
  const int q0 = (blockIdx.y * (blockDim.x >> 6) + wave) * 16 * QT;  // (1, 2 or 4 waves per workgroup: no wave talks to another)
  if (q0 >= a.nq) return;
  const f16* kbase = a.k + b * a.kbs + h * D + g * 8;
  const f16* vbase = a.vt + ((long)(b * a.heads + h) * D + c) * a.vt_ld + g * 4;
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  const f16* qrow[QT];
  bool q_ok[QT];
  f16x8 qf[QT][QREG ? KD : 1];
  f32x4 o[QT][DT];
  float mrun[QT];
  f32x4 lsum[QT];
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const int qi = q0 + u * 16 + c;
    q_ok[u] = qi < a.nq;
    qrow[u] = a.q + b * a.qbs + (long)(q_ok[u] ? qi : 0) * a.ldq + h * D + g * 8;
    if (QREG) {
#pragma unroll
      for (int kd = 0; kd < KD; ++kd) qf[u][kd] = q_ok[u] ? *(const f16x8*)(qrow[u] + kd * 32) : zero8;
    }
#pragma unroll
    for (int i = 0; i < DT; ++i) o[u][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mrun[u] = -INFINITY;
    lsum[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  int kb = 0;
  if (D <= 128) {  // 64-key tiles while they are full; the 32-key form handles the rest
    for (; kb + 64 <= (a.causal ? min(a.nkv, q0 + 16 * QT) : a.nkv); kb += 64)
      attn_tile<D, QREG, 4, QT>(a, qrow, q_ok, qf, kbase, vbase, kb, g, c, o, mrun, lsum, q0);
  }
  // causal: keys beyond this wave's last query are never visible (and key 0 always is, so the running max is
  // finite from the first tile on)
  const int kend = a.causal ? min(a.nkv, q0 + 16 * QT) : a.nkv;
  for (; kb < kend; kb += 32) attn_tile<D, QREG, 2, QT>(a, qrow, q_ok, qf, kbase, vbase, kb, g, c, o, mrun, lsum, q0);

#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const float inv = 1.0f / lsum[u][0];
    if (!q_ok[u]) continue;
    f16* orow = a.o + b * a.obs + (long)(q0 + u * 16 + c) * a.ldo + h * D + g * 4;
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      f16x4 ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = (f16)(o[u][i][r] * inv);
      *(f16x4*)(orow + i * 16) = ov;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Cross-attention with its query projection inside (CrossAttention.forward, attention.py:170-196, behind
// BasicTransformerBlock.norm2, attention.py:213): q = to_q(LayerNorm(x)) is consumed by nothing but this attention,
// and per head it is a [d x C] slice of the weight — 14 KB at the 32x32 level — against a GEMM launch of its own
// (8-10 us under replay) plus a write and a read of q.  A wave projects the QT*16 rows it owns:
//   q^T[d][row] = sum_c W'[d][c] x[row][c]      A operand = rows of W' (gamma folded in), B operand = x rows
// with the LayerNorm folded the way igemm's ln_colsum does it: q = rstd * (acc - mean * colsum(W')) + W beta, the row
// sums taken from the x fragments the lane loads anyway.  The host stores W' with its rows permuted so that the two
// 16-row output tiles of a 32-wide d chunk leave lane (g, c) with d = 32 kd + 8 g + 0..7 of row c — exactly the
// B-operand fragment of the S^T = K Q^T MFMA: no cross-lane movement, K stays as it is.
struct QProj {
  const f16* x;      // [B][nq][ldx] un-normalised rows
  const f16* w;      // [heads][D (row-permuted)][cq] fp16, LayerNorm gamma folded in
  const float* u;    // [heads * D] column sums of the fp16-rounded W' (natural d order)
  const float* bias; // [heads * D] W beta (+ the Linear's bias) (natural d order)
  long xbs;
  int ldx, cq, ln_dim;
  float eps;
};

template <int D, int QT>
__global__ __launch_bounds__(256) void attn_qproj_kernel(const AttnArgs a, const QProj p) {
  constexpr int KD = D / 32;
  constexpr int DT = D / 16;
  constexpr int KC = 7;  // x / W' chunks in flight per batch (C = 224 * n)
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int bh = blockIdx.y;
  const int b = bh / a.heads;
  const int h = bh - b * a.heads;
  const int q0 = (blockIdx.x * (blockDim.x >> 6) + wave) * 16 * QT;  // (1, 2 or 4 independent waves per workgroup)
  if (q0 >= a.nq) return;
  const f16* kbase = a.k + b * a.kbs + h * D + g * 8;
  const f16* vbase = a.vt + ((long)(b * a.heads + h) * D + c) * a.vt_ld + g * 4;

  const f16* qrow[QT];  // (unused by attn_tile with QREG = 1)
  bool q_ok[QT];
  f16x8 qf[QT][KD];
  f32x4 o[QT][DT];
  float mrun[QT];
  f32x4 lsum[QT];
  {
    f32x4 qa[QT][DT];
    float s1[QT], s2[QT];
    const f16* xr[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      const int qi = q0 + u * 16 + c;
      q_ok[u] = qi < a.nq;
      qrow[u] = nullptr;
      xr[u] = p.x + b * p.xbs + (long)(q_ok[u] ? qi : 0) * p.ldx + g * 8;
      s1[u] = s2[u] = 0.f;
#pragma unroll
      for (int i = 0; i < DT; ++i) qa[u][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const f16* wr = p.w + (long)(h * D + c) * p.cq + g * 8;
    const long wtile = (long)16 * p.cq;
    const f16x2 one2 = {(f16)1.f, (f16)1.f};
    for (int k0 = 0; k0 < p.cq; k0 += KC * 32) {
      f16x8 xv[QT][KC], wv[DT][KC];
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
        for (int u = 0; u < QT; ++u) xv[u][kc] = *(const f16x8*)(xr[u] + k0 + kc * 32);
#pragma unroll
        for (int i = 0; i < DT; ++i) wv[i][kc] = *(const f16x8*)(wr + i * wtile + k0 + kc * 32);
      }
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
        for (int u = 0; u < QT; ++u) {
#pragma unroll
          for (int hh = 0; hh < 4; ++hh) {
            const f16x2 xx = {xv[u][kc][2 * hh], xv[u][kc][2 * hh + 1]};
            s1[u] = __builtin_amdgcn_fdot2(xx, one2, s1[u], false);
            s2[u] = __builtin_amdgcn_fdot2(xx, xx, s2[u], false);
          }
#pragma unroll
          for (int i = 0; i < DT; ++i)
            qa[u][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[i][kc], xv[u][kc], qa[u][i], 0, 0, 0);
        }
      }
    }
    const float inv_dim = 1.0f / (float)p.ln_dim;
    float mean[QT], rstd[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      const float t1 = sum_over_key_groups(s1[u]), t2 = sum_over_key_groups(s2[u]);
      mean[u] = t1 * inv_dim;
      rstd[u] = rsqrtf(fmaxf(t2 * inv_dim - mean[u] * mean[u], 0.f) + p.eps);
    }
#pragma unroll
    for (int kd = 0; kd < KD; ++kd) {
      const int d0 = h * D + kd * 32 + g * 8;
      const f32x4 u0 = *(const f32x4*)(p.u + d0), u1 = *(const f32x4*)(p.u + d0 + 4);
      const f32x4 b0 = *(const f32x4*)(p.bias + d0), b1 = *(const f32x4*)(p.bias + d0 + 4);
#pragma unroll
      for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          qf[u][kd][r] = (f16)((qa[u][2 * kd][r] - mean[u] * u0[r]) * rstd[u] + b0[r]);
          qf[u][kd][4 + r] = (f16)((qa[u][2 * kd + 1][r] - mean[u] * u1[r]) * rstd[u] + b1[r]);
        }
    }
  }
#pragma unroll
  for (int u = 0; u < QT; ++u) {
#pragma unroll
    for (int i = 0; i < DT; ++i) o[u][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mrun[u] = -INFINITY;
    lsum[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  int kb = 0;
  for (; kb + 64 <= a.nkv; kb += 64) attn_tile<D, 1, 4, QT>(a, qrow, q_ok, qf, kbase, vbase, kb, g, c, o, mrun, lsum, q0);
  for (; kb < a.nkv; kb += 32) attn_tile<D, 1, 2, QT>(a, qrow, q_ok, qf, kbase, vbase, kb, g, c, o, mrun, lsum, q0);
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const float inv = 1.0f / lsum[u][0];
    if (!q_ok[u]) continue;
    f16* orow = a.o + b * a.obs + (long)(q0 + u * 16 + c) * a.ldo + h * D + g * 4;
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      f16x4 ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = (f16)(o[u][i][r] * inv);
      *(f16x4*)(orow + i * 16) = ov;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Self-attention with long key sequences (UNet level 1: 1024 / 768 keys, d = 32): the direct
// form above makes every wave pull its own K / V^T fragments through the CU's L1 — 16 cache
// lines per load instruction, 32..64 useful bytes per line, the same data four times per
// workgroup — and the level-1 launches sit at ~48 us for 8.6 GFLOP.  Here the workgroup stages each
// 64-key tile ONCE in LDS (16-byte coalesced global loads, register double buffering, one barrier
// per tile) and the four waves read their MFMA fragments from LDS:
//   K tile  [kd][64 keys][32 halfs], 16-byte chunks XOR-swizzled like the igemm tiles;
//   V^T tile [D rows][64 keys + 8 pad]  (144-byte rows: the 8-byte fragment reads spread over banks).
constexpr int ATT_VROW = 72;
__device__ __forceinline__ int att_swz(int row, int chunk) { return chunk ^ ((-(row >> 2)) & 3); }

template <int D, int QT>
__global__ __launch_bounds__(256) void attn_lds_kernel(const AttnArgs a) {
  constexpr int KD = D / 32;
  constexpr int DT = D / 16;
  __shared__ __attribute__((aligned(16))) f16 sk[2][KD * 64 * 32];
  __shared__ __attribute__((aligned(16))) f16 sv[2][D * ATT_VROW];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int bh = blockIdx.x;  // (see attn_kernel)
#ifndef UPK_NO_XCD_SAMPLE
  const int nb_ = gridDim.x / a.heads;
  const int b = bh % nb_;
  const int h = bh / nb_;
#else
  const int b = bh / a.heads;
  const int h = bh - b * a.heads;
#endif
  const int q0 = (blockIdx.y * 4 + wave) * 16 * QT;
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  bool q_ok[QT];
  f16x8 qf[QT][KD];
  f32x4 o[QT][DT];
  float mrun[QT];
  f32x4 lsum[QT];
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const int qi = q0 + u * 16 + c;
    q_ok[u] = qi < a.nq;
    const f16* qrow = a.q + b * a.qbs + (long)(q_ok[u] ? qi : 0) * a.ldq + h * D + g * 8;
#pragma unroll
    for (int kd = 0; kd < KD; ++kd) qf[u][kd] = q_ok[u] ? *(const f16x8*)(qrow + kd * 32) : zero8;
#pragma unroll
    for (int i = 0; i < DT; ++i) o[u][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mrun[u] = -INFINITY;
    lsum[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // staging assignment: K element (kd = it, key = tid>>2, chunk = tid&3); V^T element (row = (tid + 256 it)>>3, col = tid&7)
  const f16* ksrc = a.k + b * a.kbs + (long)(tid >> 2) * a.ldk + h * D + (tid & 3) * 8;
  const f16* vsrc = a.vt + ((long)(b * a.heads + h) * D + (tid >> 3)) * a.vt_ld + (tid & 7) * 8;
  const int kdst = (tid >> 2) * 32 + att_swz(tid >> 2, tid & 3) * 8;
  const int vdst = (tid >> 3) * ATT_VROW + (tid & 7) * 8;
  f16x8 kreg[KD], vreg[KD];
  auto fetch = [&](int kb) {
#pragma unroll
    for (int it = 0; it < KD; ++it) {
      kreg[it] = *(const f16x8*)(ksrc + (long)kb * a.ldk + it * 32);
      vreg[it] = *(const f16x8*)(vsrc + (long)it * 32 * a.vt_ld + kb);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int it = 0; it < KD; ++it) {
      *(f16x8*)(&sk[buf][it * 64 * 32 + kdst]) = kreg[it];
      *(f16x8*)(&sv[buf][it * 32 * ATT_VROW + vdst]) = vreg[it];
    }
  };
  const int ntiles = a.nkv >> 6;
  fetch(0);
  stash(0);
  __syncthreads();
  const float cs = a.scale_log2;
  const int kfrag = c * 32 + att_swz(c, g) * 8;
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) fetch((t + 1) << 6);
    // ---- S^T = K Q^T for 64 keys ----
    f32x4 sc[QT][4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int u = 0; u < QT; ++u) sc[u][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kd = 0; kd < KD; ++kd)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const f16x8 kf = *(const f16x8*)(&sk[buf][(kd * 64 + tt * 16) * 32 + kfrag]);
#pragma unroll
        for (int u = 0; u < QT; ++u) sc[u][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[u][kd], sc[u][tt], 0, 0, 0);
      }
    // ---- online softmax (keys 16tt + 4g + r of query c) ----
    f16x8 pf[QT][2];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      float mx = -INFINITY;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
        mx = fmaxf(mx, fmaxf(fmaxf(sc[u][tt][0], sc[u][tt][1]), fmaxf(sc[u][tt][2], sc[u][tt][3])));
      mx = max_over_key_groups(mx);
      const float mnew = fmaxf(mrun[u], mx);
      if (__builtin_amdgcn_ballot_w64(mnew != mrun[u]) != 0) {  // (see attn_tile)
        const float alpha = raw_exp2((mrun[u] - mnew) * cs);
        lsum[u] *= alpha;
#pragma unroll
        for (int i = 0; i < DT; ++i) o[u][i] *= alpha;
      }
      mrun[u] = mnew;
      const float mc = -mnew * cs;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) pf[u][tt >> 1][(tt & 1) * 4 + r] = (f16)raw_exp2(fmaf(sc[u][tt][r], cs, mc));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)  // denominators (see attn_tile)
#pragma unroll
      for (int u = 0; u < QT; ++u) lsum[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ATT_ONES, pf[u][j], lsum[u], 0, 0, 0);
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f16* vr = &sv[buf][(i * 16 + c) * ATT_VROW + 32 * j + 4 * g];
        const f16x4 va = *(const f16x4*)vr, vb = *(const f16x4*)(vr + 16);
        const f16x8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
#pragma unroll
        for (int u = 0; u < QT; ++u) o[u][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[u][j], o[u][i], 0, 0, 0);
      }
    if (t + 1 < ntiles) stash(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const float inv = 1.0f / lsum[u][0];
    if (!q_ok[u]) continue;
    f16* orow = a.o + b * a.obs + (long)(q0 + u * 16 + c) * a.ldo + h * D + g * 4;
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      f16x4 ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = (f16)(o[u][i][r] * inv);
      *(f16x4*)(orow + i * 16) = ov;
    }
  }
}

}  // namespace

// Waves per workgroup of the attention kernels whose waves are independent (no LDS, no barrier): 4, halved while the
// launch has fewer than two workgroups per CU.  The UNet's 8x8 / 4x4-level launches are 64 (sample, head) pairs x 1-4
// query groups: as 64 workgroups of 4 waves they load K / V through 64 CUs' vector-memory paths, as 256 of one wave
// through all of them (forward 3.000 -> 2.966 ms; the VAE's d = 512 mid-block attention 0.40 -> 0.28 ms).
// UPK_ATTN_WPB=n forces n (dev A/B).
static int attn_waves_per_block(const upk_ctx* ctx, int bh, int n_q, int qt) {
  static const int wpb_env = getenv("UPK_ATTN_WPB") ? atoi(getenv("UPK_ATTN_WPB")) : 0;
  if (wpb_env == 1 || wpb_env == 2 || wpb_env == 4) return wpb_env;
  int wpb = 4;
  while (wpb > 1 && (long)bh * ((n_q + 16 * qt * wpb - 1) / (16 * qt * wpb)) < 2L * ctx->num_cus) wpb >>= 1;
  return wpb;
}

// ---------------------------------------------------------------------------------------
// Wide single-head attention: the VAE's mid-block AttnBlock (model.py:150-202: one head, d = 512, 1024 / 768 tokens).
// attn_kernel<512> gives every wave of 16 queries its own pass over K and V^T through the CU's L1 — 2 MB per wave, 64
// times per sample: 284 us per decode at B = 8 for 17 GFLOP (profiles/r04_optrace_vae_32x32.txt).  Here a workgroup of
// four waves = 64 queries shares every 32-key tile through LDS:
//   * the tile arrives by LDS-DMA (global_load_lds_dwordx4: 32 requests for K, 32 for V^T, 16 per wave), double
//     buffered: tile t + 1 is on its way while the waves work on tile t, one workgroup barrier per tile;
//   * K tile [32 keys][1024 + 32 bytes] (row stride 32 mod 64: conflict-free 16-byte fragment reads), V^T tile
//     [512 rows][64 bytes] with the four 16-byte pieces of a row XOR-swizzled by (row >> 2) & 3 on the SOURCE side of
//     the DMA (the 32 lanes of an 8-byte read group then fall on 32 different 8-byte slots);
//   * per wave: Q fragments (64 registers) and the 16 x 512 output (128 registers) stay in registers, S^T / P as in
//     attn_tile (key permutation 8 g + 4 t + r -> 16 t + 4 g + r).
// Bytes: a workgroup streams K + V^T once (2 MB) for 64 queries instead of once per 16: 268 MB through the L2s per
// decode instead of 1 GB, and the 65 MFMAs a wave issues per tile stand behind LDS, not L2, latency.
template <int D>
__global__ __launch_bounds__(256) void attn_wide_kernel(const AttnArgs a) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  constexpr int KD = D / 32, DT = D / 16;
  constexpr int KROW = D * 2 + 32;         // bytes per staged K row
  constexpr int KBYTES = 32 * KROW;        // K tile
  constexpr int VBYTES = D * 64;           // V^T tile: D rows x 32 keys
  constexpr int BUF = KBYTES + VBYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int b = blockIdx.x;  // (one head: a sample per grid column; its query blocks go to one XCD when the batch is a multiple of 8)
  const int q0 = (blockIdx.y * 4 + wave) * 16;
  const f16* kg = a.k + b * a.kbs;
  const f16* vg = a.vt + (long)b * D * a.vt_ld;
  auto issue = [&](int kb, int buf) {
    char* base = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // K: row r of the tile = 64 lanes x 16 bytes (D = 512: exactly one request)
      const int r = wave + 4 * i;
      const f16* src = kg + (long)(kb + r) * a.ldk + lane * 8;
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(base + r * KROW), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < D / 64; ++i) {  // V^T: 16 rows x 64 bytes per request, pieces swizzled at the source
      const int rb = (wave + 4 * i) * 16;  // first row of this request
      const int row = rb + (lane >> 2), pc = (lane & 3) ^ ((row >> 2) & 3);
      const f16* src = vg + (long)row * a.vt_ld + kb + pc * 8;
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(base + KBYTES + rb * 64), 16, 0, 0);
    }
  };
  issue(0, 0);
  const bool q_ok = q0 + c < a.nq;
  const f16* qrow = a.q + b * a.qbs + (long)(q_ok ? q0 + c : 0) * a.ldq + g * 8;
  f16x8 qf[KD];
#pragma unroll
  for (int kd = 0; kd < KD; ++kd) qf[kd] = q_ok ? *(const f16x8*)(qrow + kd * 32) : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
  f32x4 o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 lsum = {0.f, 0.f, 0.f, 0.f};
  float mrun = -INFINITY;
  const float cs = a.scale_log2;
  const int ntiles = a.nkv >> 5;
  const int vsw = (c >> 2) & 3;  // swizzle of this lane's V^T rows (16 i + c)
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's requests for tile t have landed
    __syncthreads();                                  // ... everybody's have, and everybody is done with tile t - 1
    if (t + 1 < ntiles) issue((t + 1) << 5, buf ^ 1);
    const char* kt = smem + buf * BUF + c * KROW + g * 16;
    const char* vt = smem + buf * BUF + KBYTES + c * 64;
    f32x4 sc[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      sc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kd = 0; kd < KD; ++kd) {
        const f16x8 kf = *(const f16x8*)(kt + tt * 16 * KROW + kd * 64);
        sc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[kd], sc[tt], 0, 0, 0);
      }
    }
    float mx = fmaxf(fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3])),
                     fmaxf(fmaxf(sc[1][0], sc[1][1]), fmaxf(sc[1][2], sc[1][3])));
    mx = max_over_key_groups(mx);
    const float mnew = fmaxf(mrun, mx);
    if (__builtin_amdgcn_ballot_w64(mnew != mrun) != 0) {
      const float alpha = raw_exp2((mrun - mnew) * cs);
      lsum *= alpha;
#pragma unroll
      for (int i = 0; i < DT; ++i) o[i] *= alpha;
    }
    mrun = mnew;
    const float mc = -mnew * cs;
    f16x8 pf;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pf[tt * 4 + r] = (f16)raw_exp2(fmaf(sc[tt][r], cs, mc));
    lsum = __builtin_amdgcn_mfma_f32_16x16x32_f16(ATT_ONES, pf, lsum, 0, 0, 0);
    // V^T fragment of rows 16 i + c: keys 4 g .. 4 g + 3 (piece g >> 1, half g & 1) and 16 + 4 g .. (piece 2 + (g >> 1))
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const char* vr = vt + i * 16 * 64 + (g & 1) * 8;
      const f16x4 va = *(const f16x4*)(vr + (((g >> 1) ^ vsw) << 4));
      const f16x4 vb = *(const f16x4*)(vr + (((2 + (g >> 1)) ^ vsw) << 4));
      const f16x8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
      o[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[i], 0, 0, 0);
    }
  }
  if (!q_ok) return;
  const float inv = 1.0f / lsum[0];
  f16* orow = a.o + b * a.obs + (long)(q0 + c) * a.ldo + g * 4;
#pragma unroll
  for (int i = 0; i < DT; ++i) {
    f16x4 ov;
#pragma unroll
    for (int r = 0; r < 4; ++r) ov[r] = (f16)(o[i][r] * inv);
    *(f16x4*)(orow + i * 16) = ov;
  }
}

static int attention_impl(upk_ctx* ctx, const void* q, int ldq, long long qbs, const void* k, int ldk, long long kbs,
                          const void* vt, int vt_ld, void* out, int ldo, long long obs, int batch, int heads, int n_q,
                          int n_kv, int d, float scale, int causal, upk_stream stream_) {
  if (!ctx) return UPK_EINVAL;
  if (!q || !k || !vt || !out) return upk_fail(ctx, UPK_EINVAL, "attention: null pointer");
  if (batch <= 0 || heads <= 0 || n_q <= 0 || n_kv <= 0) return upk_fail(ctx, UPK_EINVAL, "attention: empty");
  if ((ldq & 7) || (ldk & 7) || (vt_ld & 3) || (ldo & 3) || vt_ld < ((n_kv + 31) & ~31))
    return upk_fail(ctx, UPK_EINVAL, "attention: ldq/ldk %% 8, vt_ld %% 4 and vt_ld >= round_up(n_kv,32) required");
  hipStream_t stream = (hipStream_t)stream_;
  AttnArgs a;
  a.q = (const f16*)q;
  a.k = (const f16*)k;
  a.vt = (const f16*)vt;
  a.o = (f16*)out;
  a.ldq = ldq;
  a.ldk = ldk;
  a.vt_ld = vt_ld;
  a.ldo = ldo;
  a.qbs = qbs;
  a.kbs = kbs;
  a.obs = obs;
  a.heads = heads;
  a.nq = n_q;
  a.nkv = n_kv;
  a.scale_log2 = scale * 1.4426950408889634f;
  a.causal = causal;
  if (causal && n_q != n_kv) return upk_fail(ctx, UPK_EINVAL, "attention: causal needs n_q == n_kv");
  // two 16-query groups per wave when there are enough queries to give every CU four workgroups even so (at two per CU —
  // the 32x32 level at B = 8: 512 workgroups of 128 queries — the single-group form with its 1024 workgroups hides the
  // K / V tile latency better: forward 2.890 -> 2.867 ms, -4.5 us per launch in situ, same-box A/B of UPK_ATTN_QT)
  static const int qt_env = getenv("UPK_ATTN_QT") ? atoi(getenv("UPK_ATTN_QT")) : 0;  // dev
  const int qt = qt_env ? qt_env : ((d <= 128 && (long)((n_q + 127) / 128) * batch * heads >= 4L * ctx->num_cus) ? 2 : 1);
  // grid.x = (sample, head): workgroups go to XCDs round-robin by linear index, so with batch * heads a multiple of 8
  // every query block of a (sample, head) lands on the same XCD and K / V are fetched over the fabric once, not 8 times
  dim3 grid(batch * heads, (n_q + 64 * qt - 1) / (64 * qt)), block(256);
  upk_prof_scope prof(ctx, UPK_CLS_ATTN, stream);
  // long self-attention sequences: K / V^T tiles shared through LDS (whole 64-key tiles only)
  static const bool lds_off = getenv("UPK_ATTN_DIRECT") != nullptr;
  if (!lds_off && !causal && (d == 32 || d == 64) && n_kv >= 256 && (n_kv & 63) == 0 && (vt_ld & 7) == 0) {
    if (d == 32) {
      if (qt == 2) hipLaunchKernelGGL((attn_lds_kernel<32, 2>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((attn_lds_kernel<32, 1>), grid, block, 0, stream, a);
    } else {
      if (qt == 2) hipLaunchKernelGGL((attn_lds_kernel<64, 2>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((attn_lds_kernel<64, 1>), grid, block, 0, stream, a);
    }
    return upk_check_launch(ctx, "attention_lds");
  }
  // the kernels below keep no state between waves: fewer waves per workgroup while that gives more CUs work
  {
    const int wpb = attn_waves_per_block(ctx, batch * heads, n_q, d <= 128 ? qt : 1);
    if (d <= 128) {
      grid = dim3(batch * heads, (n_q + 16 * qt * wpb - 1) / (16 * qt * wpb));
      block = dim3(64 * wpb);
    }
  }
#define UPK_ATTN(D_, QR_)                                                                         \
  if (qt == 2)                                                                                    \
    hipLaunchKernelGGL((attn_kernel<D_, QR_, (D_ <= 128 ? 2 : 1)>), grid, block, 0, stream, a);  \
  else                                                                                            \
    hipLaunchKernelGGL((attn_kernel<D_, QR_, 1>), grid, block, 0, stream, a);
  switch (d) {
    case 32: UPK_ATTN(32, 1) break;
    case 64: UPK_ATTN(64, 1) break;
    case 128: UPK_ATTN(128, 1) break;
    case 512:
      // the VAE mid-block attention on LDS-shared key tiles (attn_wide_kernel): one head, whole 32-key tiles, rows of K
      // that are exactly one 1 KiB request; anything else falls through to the direct kernel below
      // (16-byte LDS-DMA pieces and f16x8 Q loads: every row / sample stride a multiple of 8 halves, 16-byte bases)
      static const bool wide_off = getenv("UPK_ATTN_WIDE_OFF") != nullptr;  // (read once: lane threads call this concurrently)
      if (heads == 1 && !causal && (n_kv & 31) == 0 && ldk >= 512 && !wide_off &&
          ((ldk | ldq | vt_ld) & 7) == 0 && ((kbs | qbs) & 7) == 0 &&
          (((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt) & 15) == 0) {
        static unsigned long long wide_mask = 0;  // (set under upk_lds_attr_once's mutex)
        int rcw = upk_lds_attr_once(ctx, (const void*)attn_wide_kernel<512>, &wide_mask);
        if (rcw != UPK_OK) return rcw;
        const size_t lds = 2 * (32 * (512 * 2 + 32) + 512 * 64);
        hipLaunchKernelGGL((attn_wide_kernel<512>), dim3(batch, (n_q + 63) / 64), dim3(256), lds, stream, a);
        return upk_check_launch(ctx, "attention_wide");
      }
      [[fallthrough]];
    case 256: {
      // the VAE mid-block attention (one head, d = 512, 1024 tokens: model.py:180-196): every wave streams the whole K / V^T
      // of its sample through the CU's vector-memory path, so fewer waves per workgroup until every CU has work
      // (B = 8: 128 workgroups of 4 waves -> 512 of 1)
      const int wpb = attn_waves_per_block(ctx, batch * heads, n_q, 1);
      const dim3 g2(batch * heads, (n_q + 16 * wpb - 1) / (16 * wpb)), b2(64 * wpb);
      if (d == 256) hipLaunchKernelGGL((attn_kernel<256, 1, 1>), g2, b2, 0, stream, a);
      else hipLaunchKernelGGL((attn_kernel<512, 0, 1>), g2, b2, 0, stream, a);
      break;
    }
    default: return upk_fail(ctx, UPK_ESHAPE, "attention: head dim %d not in {32,64,128,256,512}", d);
  }
#undef UPK_ATTN
  return upk_check_launch(ctx, "attention");
}

extern "C" int upk_attention_f16(upk_ctx* ctx, const void* q, int ldq, long long qbs, const void* k, int ldk,
                                 long long kbs, const void* vt, int vt_ld, void* out, int ldo, long long obs,
                                 int batch, int heads, int n_q, int n_kv, int d, float scale, upk_stream stream) {
  return attention_impl(ctx, q, ldq, qbs, k, ldk, kbs, vt, vt_ld, out, ldo, obs, batch, heads, n_q, n_kv, d, scale, 0,
                        stream);
}

extern "C" int upk_attention_causal_f16(upk_ctx* ctx, const void* q, int ldq, long long qbs, const void* k, int ldk,
                                        long long kbs, const void* vt, int vt_ld, void* out, int ldo, long long obs,
                                        int batch, int heads, int n, int d, float scale, upk_stream stream) {
  return attention_impl(ctx, q, ldq, qbs, k, ldk, kbs, vt, vt_ld, out, ldo, obs, batch, heads, n, n, d, scale, 1, stream);
}

extern "C" int upk_attention_qproj_f16(upk_ctx* ctx, const void* x, int ldx, long long xbs, int cq, int ln_dim,
                                       float ln_eps, const void* wq, const float* wq_colsum, const float* wq_bias,
                                       const void* k, int ldk, long long kbs, const void* vt, int vt_ld, void* out,
                                       int ldo, long long obs, int batch, int heads, int n_q, int n_kv, int d,
                                       float scale, upk_stream stream_) {
  if (!ctx) return UPK_EINVAL;
  if (!x || !wq || !wq_colsum || !wq_bias || !k || !vt || !out) return upk_fail(ctx, UPK_EINVAL, "attention_qproj: null pointer");
  if (batch <= 0 || heads <= 0 || n_q <= 0 || n_kv <= 0) return upk_fail(ctx, UPK_EINVAL, "attention_qproj: empty");
  if ((ldx & 7) || (ldk & 7) || (vt_ld & 3) || (ldo & 3) || vt_ld < ((n_kv + 31) & ~31))
    return upk_fail(ctx, UPK_EINVAL, "attention_qproj: ldx/ldk %% 8, vt_ld %% 4 and vt_ld >= round_up(n_kv,32) required");
  if ((d != 32 && d != 64) || cq <= 0 || cq % 224 || ln_dim <= 0 || ln_dim > cq || ldx < cq)
    return upk_fail(ctx, UPK_ESHAPE, "attention_qproj: head dim %d / %d input channels outside {32,64} / 224*n", d, cq);
  hipStream_t stream = (hipStream_t)stream_;
  AttnArgs a;
  a.q = nullptr;
  a.k = (const f16*)k;
  a.vt = (const f16*)vt;
  a.o = (f16*)out;
  a.ldq = 0;
  a.ldk = ldk;
  a.vt_ld = vt_ld;
  a.ldo = ldo;
  a.qbs = 0;
  a.kbs = kbs;
  a.obs = obs;
  a.heads = heads;
  a.nq = n_q;
  a.nkv = n_kv;
  a.scale_log2 = scale * 1.4426950408889634f;
  a.causal = 0;
  QProj p;
  p.x = (const f16*)x;
  p.w = (const f16*)wq;
  p.u = wq_colsum;
  p.bias = wq_bias;
  p.xbs = xbs;
  p.ldx = ldx;
  p.cq = cq;
  p.ln_dim = ln_dim;
  p.eps = ln_eps;
  upk_prof_scope prof(ctx, UPK_CLS_ATTN, stream);
  const int qt = d == 32 ? 2 : 1;  // (d = 64: two query groups per wave measured slower: 21.5 vs 17.8 us at 256 x 8 heads)
  const int wpb = attn_waves_per_block(ctx, batch * heads, n_q, qt);
  const dim3 grid((n_q + 16 * qt * wpb - 1) / (16 * qt * wpb), batch * heads), block(64 * wpb);
  if (d == 32) hipLaunchKernelGGL((attn_qproj_kernel<32, 2>), grid, block, 0, stream, a, p);
  else hipLaunchKernelGGL((attn_qproj_kernel<64, 1>), grid, block, 0, stream, a, p);
  return upk_check_launch(ctx, "attention_qproj");
}

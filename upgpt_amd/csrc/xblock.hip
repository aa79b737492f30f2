// The cross-attention half of a BasicTransformerBlock (attention.py:257-260) as ONE launch:
//
//     t1 = attn1.to_out(a1) + t0                                  [M, C]   (a1 = self-attention output, heads side by side)
//     q  = to_q(LayerNorm2(t1))                                   [M, heads * d]
//     a2 = softmax(q K_ctx^T * scale) V_ctx   per head            [M, heads * d]   (K / V^T of the context precomputed)
//     t2 = attn2.to_out(a2) + t1                                  [M, C]
//
// Four launches in the unfused form (to_out GEMM, LayerNorm+to_q GEMM, attention, to_out GEMM), each of them a few
// microseconds of work behind ~5 us of fixed cost at the 32x32 / 16x16 levels.  A workgroup owns BM rows of one sample
// and 8 waves = 8 heads:
//
//   G1  waves 0..6 own C / 7 columns each; A = a1 tile in LDS (LDS-DMA), weights from L2 into a register ring
//       (astat.hip's scheme, no barrier in the K loop); epilogue + bias + t0 -> t1 tile in LDS (fp16)
//   LN  row statistics of the resident t1 tile
//   G2  wave h owns head h's d query columns: A = t1 tile, folded LayerNorm in the epilogue; the queries never leave the
//       wave's registers: the accumulator layout (lane (c, g): row c, columns 16 j + 4 g .. + 3) IS the MFMA operand of
//       the score product under a permutation of the contracted index, and K is read with the same permutation
//   XA  scores over <= 96 context keys, softmax, P V (attention.hip's key permutation: P stays lane local too),
//       a2 -> LDS over the a1 tile
//   G3  like G1 with A = a2 tile, + bias + t1 (from LDS) -> t2 in HBM
//
// t1, q and a2 never exist in HBM.  The weights of the next stage are requested while the current one computes (the ring of
// G1 is refilled with G3's weights as it drains; G2's ring, the K and V^T fragments are in flight from the start).
#include "igemm_common.h"

namespace upkd {
namespace {

constexpr int XB_NW = 8;
constexpr int XB_KEYS = 96;  // context keys covered (6 fragments of 16)

struct XbArgs {
  const f16* a1;    // [M, lda]
  const f16* t0;    // [M, ldt0]
  const f16* w1;    // packed [hd / 32][C][32]
  const f16* wq;    // packed [C / 32][hd][32], LayerNorm affine folded in
  const f16* w3;    // packed [hd / 32][C][32]
  const f16* kc;    // [B * nkv, ldk]
  const f16* vt;    // [B, heads, d, vt_ld]
  const f16* zero;
  const float* vec;  // [b1 (C) | uq (hd) | bq (hd) | b3 (C)], padded to a multiple of 256 floats
  f16* y;           // t2 [M, ldy]
  int lda, ldt0, ldy, ldk, vt_ld, M, hw, nkv, vec_pieces;
  float ln_inv_dim, ln_eps, scale_log2;
  unsigned long long* dbg;
};
#define XB_PIN(v) asm volatile("" ::"s"(v))

__device__ __forceinline__ float xb_max4(float x) {
  unsigned u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u = __float_as_uint(fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])));
  const auto b = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float xb_sum4(float x) {
  unsigned u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u = __float_as_uint(__uint_as_float(a[0]) + __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Workgroup barrier for LDS traffic only: __syncthreads() carries a release fence, which on gfx9 waits for vmcnt(0) — every
// weight / K / V^T fragment requested ahead for the later stages (the in-kernel stamps showed 2-7k cycles per barrier).
__device__ __forceinline__ void xb_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// one ring block: PF chunks of A (LDS) against the ring, refilled from `sb` behind their MFMAs when REFILL
template <int MI, int NI, int PF, int BM, bool REFILL>
__device__ __forceinline__ void xb_block(f32x4 (&acc)[MI][NI], f16x8 (&ring)[PF][NI], const char* sb, unsigned ks,
                                         unsigned loff, const char* la) {
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    f16x8 fc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) fc[i] = *(const f16x8*)(la + (u * BM + i * 16) * 64);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[u][j], fc[i], acc[i][j], 0, 0, 0);
    if (REFILL) {
#pragma unroll
      for (int j = 0; j < NI; ++j) ring[u][j] = *(const f16x8*)(sb + (unsigned)u * ks + j * 1024 + loff);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// acc += A (LDS tile, NCH chunks) x W (this wave's NI fragments of 16 columns); the ring holds the first PF chunks of
// `wb` on entry and is empty on return (the last block is peeled: no refills, no dead loads)
template <int MI, int NI, int PF, int NCH, int BM>
__device__ __forceinline__ void xb_gemm(f32x4 (&acc)[MI][NI], f16x8 (&ring)[PF][NI], const char* wb, unsigned ks,
                                        unsigned loff, const char* ldsA) {
  static_assert(NCH % PF == 0, "ring depth must divide the chunk count");
#pragma unroll 1
  for (int kc0 = 0; kc0 < NCH - PF; kc0 += PF)
    xb_block<MI, NI, PF, BM, true>(acc, ring, wb + (size_t)((unsigned)(kc0 + PF) * ks), ks, loff,
                                   ldsA + (unsigned)(kc0 * BM) * 64u);
  xb_block<MI, NI, PF, BM, false>(acc, ring, wb, ks, loff, ldsA + (unsigned)((NCH - PF) * BM) * 64u);
}

template <int NI, int PF>
__device__ __forceinline__ void xb_fill(f16x8 (&ring)[PF][NI], const char* wb, unsigned ks, unsigned loff) {
#pragma unroll
  for (int u = 0; u < PF; ++u)
#pragma unroll
    for (int j = 0; j < NI; ++j) ring[u][j] = *(const f16x8*)(wb + (unsigned)u * ks + j * 1024 + loff);
}

template <int MI, int C32, int DP, int PF1, int PFQ>
__global__ __launch_bounds__(512) void xblock_kernel(const XbArgs s) {
  constexpr int NW = XB_NW, BM = MI * 16;
  constexpr int C = C32 * 32, HD = NW * DP, NCHD = HD / 32;
  static_assert(C32 % 7 == 0, "C = 7 waves x NI1 fragments");
  constexpr int NI1 = C32 * 2 / 7;  // 16-column fragments per wave over N = C (7 waves)
  constexpr int NIQ = DP / 16;      // ... over one head's query columns
  constexpr int KD = DP / 32;       // 32-deep chunks of a head
  constexpr int NKF = XB_KEYS / 16, NKC = XB_KEYS / 32;
  extern __shared__ __attribute__((aligned(16))) f16 smem[];
  XB_PIN(s.a1); XB_PIN(s.t0); XB_PIN(s.w1); XB_PIN(s.wq); XB_PIN(s.w3); XB_PIN(s.kc); XB_PIN(s.vt); XB_PIN(s.zero);
  XB_PIN(s.vec); XB_PIN(s.y); XB_PIN(s.lda); XB_PIN(s.ldt0); XB_PIN(s.ldy); XB_PIN(s.ldk); XB_PIN(s.vt_ld); XB_PIN(s.M);
  XB_PIN(s.hw); XB_PIN(s.nkv); XB_PIN(s.vec_pieces);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4, lc = lane & 15;
#ifdef UPK_TIMELINE
  const bool tl = s.dbg && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && (wave == 0 || wave == 4);
  unsigned long long* tlp = s.dbg + (blockIdx.x == 0 ? 0 : 64) + (wave == 0 ? 0 : 32);
#define STAMP(i) do { if (tl && lane == 0 && (i) < 32) tlp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
  STAMP(0);
  const int m0 = upk_xcd_tile(blockIdx.x, gridDim.x) * BM;
  const int b = m0 / s.hw;  // (hw % BM == 0: a workgroup never straddles two samples)
  f16* const aT = smem;                                // [NCHD][BM][32]  a1, later a2
  f16* const tT = smem + NCHD * BM * 32;               // [C32][BM][32]   t1
  float* const st = (float*)(tT + C32 * BM * 32);      // [BM][2]
  float* const bl = st + BM * 2;                       // vec
  const float* const bl_b1 = bl;
  const float* const bl_uq = bl + C;
  const float* const bl_bq = bl + C + HD;
  const float* const bl_b3 = bl + C + 2 * HD;

  // ---- 0. a1 tile and the epilogue vectors by LDS-DMA
  {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int r16 = lane >> 2;
    const int chd = (lane & 3) ^ ((-(lane >> 4)) & 3);
    const f16* zsrc = s.zero + (lane & 3) * 8;
    for (int idx = wave; idx < NCHD * MI; idx += NW) {
      const int kc = idx / MI, rg = idx - kc * MI;
      const int m = m0 + rg * 16 + r16;
      const f16* src = m < s.M ? s.a1 + (long)m * s.lda + kc * 32 + chd * 8 : zsrc;
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(aT + (kc * BM + rg * 16) * 32), 16, 0, 0);
    }
    for (int idx = wave; idx < s.vec_pieces; idx += NW)
      __builtin_amdgcn_global_load_lds((glb_ptr)(s.vec + idx * 256 + lane * 4), (lds_ptr)(bl + idx * 256), 16, 0, 0);
  }

  // ---- weight streams
  const bool act1 = wave < 7;
  const int col1 = wave * NI1 * 16;  // first of this wave's columns of C
  const int colq = wave * DP;        // ... of the head dimension
  const unsigned ks1 = (unsigned)C * 64u, ksq = (unsigned)HD * 64u;
  const unsigned loff = (unsigned)(lc * 32 + lg * 8) * 2u;
  // (wave 7 has no columns of C: it runs G1 / G3 on wave 6's and drops the results — ONE instruction stream for all
  // waves, so that the compiler's vmcnt bookkeeping stays exact: behind a branch around loads it falls back to vmcnt(0))
  const int colw = act1 ? col1 : 6 * NI1 * 16;
  const char* const wb1 = (const char*)s.w1 + (size_t)colw * 64;
  const char* const wb3 = (const char*)s.w3 + (size_t)colw * 64;
  const char* const wbq = (const char*)s.wq + (size_t)colq * 64;
  f16x8 ring1[PF1][NI1];
  __builtin_amdgcn_sched_barrier(0);
  xb_fill<NI1, PF1>(ring1, wb1, ks1, loff);
  __builtin_amdgcn_sched_barrier(0);
  STAMP(1);
  // the tile DMAs were this wave's FIRST vector-memory operations: they have landed once at most the ring fill behind them
  // is outstanding (counted wait: the fragments requested below for the later stages stay in flight across every barrier)
  // (the builtin, not inline asm: the compiler's own counter model sees it.  lgkmcnt(0) with it: an LDS-DMA is a FLAT
  // instruction with two address spaces to that model, and while one is "pending" on EITHER counter every later wait it
  // inserts is vmcnt(0) lgkmcnt(0) — the whole prefetch drained in front of G1's first MFMA)
  static_assert(PF1 * NI1 < 64, "vmcnt range");
  __builtin_amdgcn_s_waitcnt(((PF1 * NI1) & 15) | (((PF1 * NI1) >> 4) << 14) | 0x0070);
  __builtin_amdgcn_s_barrier();
  STAMP(2);
  f16x4 rr[MI][NI1];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI1; ++j) {
      const unsigned m = (unsigned)min(m0 + i * 16 + lc, s.M - 1);
      rr[i][j] = *(const f16x4*)(s.t0 + m * (unsigned)s.ldt0 + (unsigned)(colw + j * 16 + lg * 4));
    }
  f16x8 ringq[PFQ][NIQ];
  xb_fill<NIQ, PFQ>(ringq, wbq, ksq, loff);
  __builtin_amdgcn_sched_barrier(0);

  const unsigned la = (unsigned)(lc * 32 + lds_swz(lc, lg) * 8) * 2u;
  // this lane's 4 columns n .. n + 3 of a [chunks][BM][32] tile: chunk n / 32, 16-byte piece (n & 31) / 8 (swizzled), half
  auto tile_at = [&](f16* T, int row, int n) -> f16* {
    return T + ((n >> 5) * BM + row) * 32 + lds_swz(row & 15, (n & 31) >> 3) * 8 + (n & 7);
  };

  // ---- G1: t1 = a1 W1^T + b1 + t0 -> LDS
  {
    f32x4 acc[MI][NI1];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI1; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    xb_gemm<MI, NI1, PF1, NCHD, BM>(acc, ring1, wb1, ks1, loff, (const char*)aT + la);
#pragma unroll
    for (int j = 0; j < NI1; ++j) {
      const int n = colw + j * 16 + lg * 4;
      const f32x4 bv = *(const f32x4*)(bl_b1 + n);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        f16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (f16)(acc[i][j][k] + bv[k] + (float)rr[i][j][k]);
        if (act1) *(f16x4*)tile_at(tT, i * 16 + lc, n) = o;
      }
    }
  }
  STAMP(3);
  xb_lds_barrier();  // (t1 complete; a1 tile free)
  STAMP(4);
  // context keys of head `wave`, permuted along d: MFMA k slot (g, e) <-> d = 32 c + (e < 4 ? 4 g + e : 16 + 4 g + e - 4)
  f16x8 kf[NKF][KD];
  {
    const f16* kb = s.kc + (long)b * s.nkv * s.ldk + colq + lg * 4;
#pragma unroll
    for (int t = 0; t < NKF; ++t) {
      const int key = t * 16 + lc;
      const f16* kr = kb + (long)(key < s.nkv ? key : 0) * s.ldk;
#pragma unroll
      for (int c = 0; c < KD; ++c) {
        const f16x4 lo = *(const f16x4*)(kr + 32 * c), hi = *(const f16x4*)(kr + 32 * c + 16);
        kf[t][c] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- LayerNorm statistics of the t1 rows
  {
    constexpr int LPR = 512 / BM;
    const int row = tid / LPR, part = tid - row * LPR;
    float s1 = 0.f, s2 = 0.f;
    const f16x2 one2 = {(f16)1.f, (f16)1.f};
    for (int q = part; q < C32 * 4; q += LPR) {
      const f16x8 v = *(const f16x8*)(tT + ((q >> 2) * BM + row) * 32 + (q & 3) * 8);
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const f16x2 xx = {v[2 * h], v[2 * h + 1]};
        s1 = __builtin_amdgcn_fdot2(xx, one2, s1, false);
        s2 = __builtin_amdgcn_fdot2(xx, xx, s2, false);
      }
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) {
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    if (part == 0) {
      const float mu = s1 * s.ln_inv_dim;
      st[2 * row] = mu;
      st[2 * row + 1] = rsqrtf(fmaxf(s2 * s.ln_inv_dim - mu * mu, 0.f) + s.ln_eps);
    }
  }
  xb_lds_barrier();
  STAMP(5);

  // ---- G2: q = LayerNorm(t1) Wq^T (head `wave`), straight into the score product's operand
  f16x8 qv[MI][KD];
  {
    f32x4 acc[MI][NIQ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NIQ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    xb_gemm<MI, NIQ, PFQ, C32, BM>(acc, ringq, wbq, ksq, loff, (const char*)tT + la);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const f32x2 mr = *(const f32x2*)(st + 2 * (i * 16 + lc));
#pragma unroll
      for (int j = 0; j < NIQ; ++j) {
        const int n = colq + j * 16 + lg * 4;
        const f32x4 v = (acc[i][j] - mr[0] * *(const f32x4*)(bl_uq + n)) * mr[1] + *(const f32x4*)(bl_bq + n);
#pragma unroll
        for (int k = 0; k < 4; ++k) qv[i][j >> 1][(j & 1) * 4 + k] = (f16)v[k];
      }
    }
  }
  STAMP(6);
  // (requested here, behind G2: the registers of its ring are free, and the scores + softmax cover the latency)
  // V^T of head `wave` with attention.hip's key permutation (k slot (g, e) <-> key 32 j + (e < 4 ? 4 g + e : 16 + 4 g + e - 4))
  f16x8 vf[NIQ][NKC];
  {
    const f16* vb = s.vt + ((long)(b * NW + wave) * DP + lc) * s.vt_ld + lg * 4;
#pragma unroll
    for (int i = 0; i < NIQ; ++i)
#pragma unroll
      for (int j = 0; j < NKC; ++j) {
        const f16* vr = vb + (long)i * 16 * s.vt_ld + 32 * j;
        const f16x4 lo = *(const f16x4*)vr, hi = *(const f16x4*)(vr + 16);
        vf[i][j] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
  }
  xb_fill<NI1, PF1>(ring1, wb3, ks1, loff);
  __builtin_amdgcn_sched_barrier(0);

  // ---- XA: softmax(q K^T scale) V over the context keys of head `wave`
  {
    const float cs = s.scale_log2;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      f32x4 sc[NKF];
#pragma unroll
      for (int t = 0; t < NKF; ++t) {
        sc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KD; ++c) sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[t][c], qv[i][c], sc[t], 0, 0, 0);
      }
      // lane (g, c) holds keys 16 t + 4 g + r of query row c; only the fragments that reach past n_kv need the mask
      // (wave-uniform test per fragment: for 87 keys that is the last one of six)
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < NKF; ++t) {
        if (16 * t + 16 > s.nkv) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * t + 4 * lg + r >= s.nkv) sc[t][r] = -INFINITY;
        }
        mx = fmaxf(mx, fmaxf(fmaxf(sc[t][0], sc[t][1]), fmaxf(sc[t][2], sc[t][3])));
      }
      mx = xb_max4(mx);
      const float mc = -mx * cs;
      float l = 0.f;
      f16x8 pf[NKC];
#pragma unroll
      for (int t = 0; t < NKF; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(sc[t][r], cs, mc));
          pf[t >> 1][(t & 1) * 4 + r] = (f16)p;
          l += p;
        }
      const float inv = __builtin_amdgcn_rcpf(xb_sum4(l));
#pragma unroll
      for (int d = 0; d < NIQ; ++d) {
        f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NKC; ++j) o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[d][j], pf[j], o, 0, 0, 0);
        f16x4 h;
#pragma unroll
        for (int k = 0; k < 4; ++k) h[k] = (f16)(o[k] * inv);
        *(f16x4*)tile_at(aT, i * 16 + lc, colq + d * 16 + lg * 4) = h;
      }
    }
  }
  STAMP(7);
  xb_lds_barrier();  // (a2 complete)
  STAMP(8);

  // ---- G3: t2 = a2 W3^T + b3 + t1 -> HBM
  {
    f32x4 acc[MI][NI1];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI1; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    xb_gemm<MI, NI1, PF1, NCHD, BM>(acc, ring1, wb3, ks1, loff, (const char*)aT + la);
    STAMP(9);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + i * 16 + lc;
      const bool ok = m < s.M && act1;
      f16* yrow = s.y + (unsigned)(ok ? m : 0) * (unsigned)s.ldy;
#pragma unroll
      for (int j = 0; j < NI1; ++j) {
        const int n = colw + j * 16 + lg * 4;
        const f32x4 bv = *(const f32x4*)(bl_b3 + n);
        const f16x4 r = *(const f16x4*)tile_at(tT, i * 16 + lc, n);
        f16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (f16)(acc[i][j][k] + bv[k] + (float)r[k]);
        if (ok) *(f16x4*)(yrow + n) = o;
      }
    }
  }
#ifdef UPK_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(10);
#endif
#undef STAMP
}


// ------------------------------------------------------------------------------------------------------------------
// The head of a SpatialTransformer (attention.py:330-333, 257): proj_in -> norm1 -> to_q | to_k | to_v of the
// self-attention, one launch:  t0 = xn W_in^T + b_in  (xn = the GroupNorm output);  q | k | v = LayerNorm(t0) W_qkv^T.
// Same scheme as above: G1 on 7 waves (C / 7 columns each, + bias -> t0 tile in LDS and in HBM: the block's residual
// stream), row statistics of the tile, then three passes of 8 waves x d columns — pass 0 = q, 1 = k, 2 = v of head
// `wave` — with the weight ring refilled across the passes; v goes out transposed ([B, heads, d, vt_ld], what
// upk_attention_f16 reads).
struct HbArgs {
  const f16* x;      // [M, ldx]
  const f16* w1;     // packed [C / 32][C][32]
  const f16* w2;     // packed [C / 32][3 hd][32], LayerNorm affine folded in
  const f16* zero;
  const float* vec;  // [b1 (C) | u2 (3 hd) | b2 (3 hd)], padded to a multiple of 256 floats
  f16* t0;           // [M, ldt0]
  f16* qk;           // [M, ldqk]: q | k
  f16* vt;           // [B, heads, d, vt_ld]
  int ldx, ldt0, ldqk, vt_ld, M, hw, vec_pieces;
  float ln_inv_dim, ln_eps;
  // GroupNorm of the input folded in (GNF): x is un-normalised, the statistics come from the per-(row block, channel)
  // partials its producer's epilogue left ([B][gn_nblk][2][gn_ld], include/upk.h gn_stats_ws mode 2)
  const float* gn_part;
  const float* gn_gamma;
  const float* gn_beta;
  int gn_nblk, gn_ld, gn_cpg;
  float gn_eps;
  unsigned long long* dbg;
};

template <int MI, int C32, int DP, bool GNF>
__global__ __launch_bounds__(512) void hblock_kernel(const HbArgs s) {
  constexpr int NW = XB_NW, BM = MI * 16;
  constexpr int C = C32 * 32, HD = NW * DP, N2 = 3 * HD;
  static_assert(C32 % 7 == 0, "C = 7 waves x NI1 fragments");
  constexpr int NI1 = C32 * 2 / 7;
  constexpr int NIQ = DP / 16;
  constexpr int PF = 7;
  static_assert(C32 % PF == 0, "ring depth");
  extern __shared__ __attribute__((aligned(16))) f16 smem[];
  XB_PIN(s.x); XB_PIN(s.w1); XB_PIN(s.w2); XB_PIN(s.zero); XB_PIN(s.vec); XB_PIN(s.t0); XB_PIN(s.qk); XB_PIN(s.vt);
  XB_PIN(s.ldx); XB_PIN(s.ldt0); XB_PIN(s.ldqk); XB_PIN(s.vt_ld); XB_PIN(s.M); XB_PIN(s.hw); XB_PIN(s.vec_pieces);
  if (GNF) { XB_PIN(s.gn_part); XB_PIN(s.gn_gamma); XB_PIN(s.gn_beta); XB_PIN(s.gn_nblk); XB_PIN(s.gn_ld); XB_PIN(s.gn_cpg); }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4, lc = lane & 15;
#ifdef UPK_TIMELINE
  const bool tl = s.dbg && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && (wave == 0 || wave == 4);
  unsigned long long* tlp = s.dbg + (blockIdx.x == 0 ? 0 : 64) + (wave == 0 ? 0 : 32);
#define STAMP(i) do { if (tl && lane == 0 && (i) < 32) tlp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
  STAMP(0);
  const int m0 = upk_xcd_tile(blockIdx.x, gridDim.x) * BM;
  const int b = m0 / s.hw;
  const int tok0 = m0 - b * s.hw;
  f16* const aT = smem;                            // [C32][BM][32]  xn (x until the GroupNorm is applied in place)
  f16* const tT = smem + C32 * BM * 32;            // [C32][BM][32]  t0
  float* const st = (float*)(tT + C32 * BM * 32);  // [BM][2]
  float* const bl = st + BM * 2;
  const float* const bl_b1 = bl;
  const float* const bl_u2 = bl + C;
  const float* const bl_b2 = bl + C + N2;
  float* const tab = bl + s.vec_pieces * 256;      // GNF: [2][C] channel sums, then scale | shift
  double* const gsum = (double*)(tab + 2 * C);     // GNF: [groups][2]
  float* const gst = (float*)(gsum + 2 * UPK_GN_GROUPS_MAX);  // GNF: mean[groups], rstd[groups]
  {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int r16 = lane >> 2;
    const int chd = (lane & 3) ^ ((-(lane >> 4)) & 3);
    const f16* zsrc = s.zero + (lane & 3) * 8;
    for (int idx = wave; idx < C32 * MI; idx += NW) {
      const int kc = idx / MI, rg = idx - kc * MI;
      const int m = m0 + rg * 16 + r16;
      const f16* src = m < s.M ? s.x + (long)m * s.ldx + kc * 32 + chd * 8 : zsrc;
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(aT + (kc * BM + rg * 16) * 32), 16, 0, 0);
    }
    for (int idx = wave; idx < s.vec_pieces; idx += NW)
      __builtin_amdgcn_global_load_lds((glb_ptr)(s.vec + idx * 256 + lane * 4), (lds_ptr)(bl + idx * 256), 16, 0, 0);
  }
  // GNF: this thread's (sum | sum of squares, channel) column of the producer's partials, blocks added in index order
  // (norm.hip gn_apply_kernel's arithmetic: fp32 over the blocks, fp64 over a group's channels), and its gamma / beta
  float gn_acc = 0.f, gn_g = 0.f, gn_b = 0.f;
  if (GNF) {
    if (tid < 2 * C) {
      const int which = tid >= C ? 1 : 0;
      const float* src = s.gn_part + (long)b * s.gn_nblk * 2 * s.gn_ld + which * s.gn_ld + (tid - which * C);
#pragma unroll 8
      for (int k = 0; k < s.gn_nblk; ++k) gn_acc += src[(long)k * 2 * s.gn_ld];
    }
    if (tid < C) gn_g = s.gn_gamma[tid], gn_b = s.gn_beta[tid];
  }
  const bool act1 = wave < 7;
  const int colw = (act1 ? wave : 6) * NI1 * 16;  // (wave 7: wave 6's columns, results dropped — one instruction stream)
  const int colq = wave * DP;
  const unsigned ks1 = (unsigned)C * 64u, ks2 = (unsigned)N2 * 64u;
  const unsigned loff = (unsigned)(lc * 32 + lg * 8) * 2u;
  const char* const wb1 = (const char*)s.w1 + (size_t)colw * 64;
  const char* const wb2 = (const char*)s.w2 + (size_t)colq * 64;
  f16x8 ring1[PF][NI1];
  __builtin_amdgcn_sched_barrier(0);
  xb_fill<NI1, PF>(ring1, wb1, ks1, loff);
  __builtin_amdgcn_sched_barrier(0);
  STAMP(1);
  static_assert(PF * NI1 < 64, "vmcnt range");
  __builtin_amdgcn_s_waitcnt(((PF * NI1) & 15) | (((PF * NI1) >> 4) << 14) | 0x0070);  // (see xblock_kernel)
  __builtin_amdgcn_s_barrier();
  STAMP(2);
  f16x8 ring2[PF][NIQ];
  xb_fill<NIQ, PF>(ring2, wb2, ks2, loff);
  __builtin_amdgcn_sched_barrier(0);
  if (GNF) {
    if (tid < 2 * C) tab[tid] = gn_acc;
    xb_lds_barrier();
    const int groups = C / s.gn_cpg;
    if (tid < groups * 2) {
      const int g = tid >> 1, which = tid & 1;
      double acc = 0.0;
      for (int e = 0; e < s.gn_cpg; ++e) acc += (double)tab[which * C + g * s.gn_cpg + e];
      gsum[tid] = acc;
    }
    xb_lds_barrier();
    if (tid < groups) {
      const double n = (double)s.hw * s.gn_cpg;
      const double mean = gsum[tid * 2] / n;
      double var = gsum[tid * 2 + 1] / n - mean * mean;
      if (var < 0.0) var = 0.0;
      gst[tid] = (float)mean;
      gst[UPK_GN_GROUPS_MAX + tid] = (float)(1.0 / sqrt(var + (double)s.gn_eps));
    }
    xb_lds_barrier();
    if (tid < C) {
      const int g = tid / s.gn_cpg;
      const float sc = gst[UPK_GN_GROUPS_MAX + g] * gn_g;
      tab[tid] = sc;
      tab[C + tid] = gn_b - gst[g] * sc;
    }
    xb_lds_barrier();
    // the tile in place: y = fp16(x * scale + shift), the value the GroupNorm launch would have stored
    for (int v = tid; v < C32 * BM * 4; v += 512) {
      const int kc = v / (BM * 4), rem = v - kc * (BM * 4);
      const int row = rem >> 2, slot = rem & 3;
      const int ch = kc * 32 + (slot ^ ((-((row & 15) >> 2)) & 3)) * 8;  // (the piece that the swizzle put in this slot)
      f16x8* p = (f16x8*)(aT + (kc * BM + row) * 32 + slot * 8);
      const f16x8 xv = *p;
      const f32x4 sc0 = *(const f32x4*)(tab + ch), sc1 = *(const f32x4*)(tab + ch + 4);
      const f32x4 sh0 = *(const f32x4*)(tab + C + ch), sh1 = *(const f32x4*)(tab + C + ch + 4);
      f16x8 yv;
#pragma unroll
      for (int j = 0; j < 8; ++j) yv[j] = (f16)((float)xv[j] * (j < 4 ? sc0[j] : sc1[j - 4]) + (j < 4 ? sh0[j] : sh1[j - 4]));
      *p = yv;
    }
    xb_lds_barrier();
  }

  const unsigned la = (unsigned)(lc * 32 + lds_swz(lc, lg) * 8) * 2u;
  auto tile_at = [&](f16* T, int row, int n) -> f16* {
    return T + ((n >> 5) * BM + row) * 32 + lds_swz(row & 15, (n & 31) >> 3) * 8 + (n & 7);
  };

  // ---- G1: t0 = xn W1^T + b1 -> LDS and HBM
  {
    f32x4 acc[MI][NI1];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI1; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    xb_gemm<MI, NI1, PF, C32, BM>(acc, ring1, wb1, ks1, loff, (const char*)aT + la);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + i * 16 + lc;
      const bool ok = m < s.M && act1;
      f16* trow = s.t0 + (unsigned)(m < s.M ? m : 0) * (unsigned)s.ldt0;
#pragma unroll
      for (int j = 0; j < NI1; ++j) {
        const int n = colw + j * 16 + lg * 4;
        const f32x4 bv = *(const f32x4*)(bl_b1 + n);
        f16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (f16)(acc[i][j][k] + bv[k]);
        if (act1) *(f16x4*)tile_at(tT, i * 16 + lc, n) = o;
        if (ok) *(f16x4*)(trow + n) = o;
      }
    }
  }
  STAMP(3);
  xb_lds_barrier();
  STAMP(4);
  {
    constexpr int LPR = 512 / BM;
    const int row = tid / LPR, part = tid - row * LPR;
    float s1 = 0.f, s2 = 0.f;
    const f16x2 one2 = {(f16)1.f, (f16)1.f};
    for (int q = part; q < C32 * 4; q += LPR) {
      const f16x8 v = *(const f16x8*)(tT + ((q >> 2) * BM + row) * 32 + (q & 3) * 8);
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const f16x2 xx = {v[2 * h], v[2 * h + 1]};
        s1 = __builtin_amdgcn_fdot2(xx, one2, s1, false);
        s2 = __builtin_amdgcn_fdot2(xx, xx, s2, false);
      }
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) {
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    if (part == 0) {
      const float mu = s1 * s.ln_inv_dim;
      st[2 * row] = mu;
      st[2 * row + 1] = rsqrtf(fmaxf(s2 * s.ln_inv_dim - mu * mu, 0.f) + s.ln_eps);
    }
  }
  xb_lds_barrier();
  STAMP(5);

  // ---- G2: three passes (q, k, v of head `wave`), the ring refilled with the next pass's weights as it drains
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    f32x4 acc[MI][NIQ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NIQ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    static_assert(C32 == PF, "one ring block per pass");
    if (p < 2) xb_block<MI, NIQ, PF, BM, true>(acc, ring2, wb2 + (size_t)(p + 1) * HD * 64, ks2, loff, (const char*)tT + la);
    else xb_block<MI, NIQ, PF, BM, false>(acc, ring2, wb2, ks2, loff, (const char*)tT + la);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const f32x2 mr = *(const f32x2*)(st + 2 * (i * 16 + lc));
      const int m = m0 + i * 16 + lc;
      const bool ok = m < s.M;
#pragma unroll
      for (int j = 0; j < NIQ; ++j) {
        const int n = p * HD + colq + j * 16 + lg * 4;
        const f32x4 v = (acc[i][j] - mr[0] * *(const f32x4*)(bl_u2 + n)) * mr[1] + *(const f32x4*)(bl_b2 + n);
        if (p < 2) {
          f16x4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = (f16)v[k];
          if (ok) *(f16x4*)(s.qk + (unsigned)m * (unsigned)s.ldqk + n) = o;
        } else if (ok) {
          f16* dst = s.vt + ((long)(b * NW + wave) * DP + j * 16 + lg * 4) * s.vt_ld + tok0 + i * 16 + lc;
#pragma unroll
          for (int k = 0; k < 4; ++k) dst[(unsigned)k * (unsigned)s.vt_ld] = (f16)v[k];
        }
      }
    }
    STAMP(6 + p);
  }
#ifdef UPK_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(10);
#endif
#undef STAMP
}

}  // namespace
}  // namespace upkd

using namespace upkd;

namespace {
struct XbCfg {
  int c, dp, bm;
  void (*fn)(const XbArgs);
  unsigned long long attr_done;  // (devices on which the LDS attribute is set)
};
XbCfg g_xb[] = {
    {224, 32, 32, xblock_kernel<2, 7, 32, 8, 7>, 0},
    {224, 32, 16, xblock_kernel<1, 7, 32, 8, 7>, 0},
    {448, 64, 32, xblock_kernel<2, 14, 64, 4, 2>, 0},
    {448, 64, 16, xblock_kernel<1, 14, 64, 4, 7>, 0},
};
XbCfg* xb_find(const upk_xblock_desc* d) {
  const int bm = d->rows_per_wg > 0 ? d->rows_per_wg : 32;
  for (auto& c : g_xb)
    if (c.c == d->c && c.dp == d->d && c.bm == bm) return &c;
  return nullptr;
}
}  // namespace

extern "C" int upk_cross_block_supported(upk_ctx* ctx, const upk_xblock_desc* d) {
  if (!ctx || !d) return 0;
  if (d->heads != XB_NW || d->n_kv <= 0 || d->n_kv > XB_KEYS || d->vt_ld < XB_KEYS) return 0;
  if (!xb_find(d)) return 0;
  const int bm = d->rows_per_wg > 0 ? d->rows_per_wg : 32;
  if (d->hw <= 0 || d->hw % bm || d->m % d->hw) return 0;
  if ((d->lda & 7) || (d->ld_t0 & 3) || (d->ldy & 3) || (d->ldk & 3) || (d->vt_ld & 3)) return 0;
  return 1;
}

extern "C" int upk_cross_block_f16(upk_ctx* ctx, const upk_xblock_desc* d, upk_stream stream_) {
  if (!ctx || !d) return UPK_EINVAL;
  if (!d->a1 || !d->t0 || !d->w_out1 || !d->w_q || !d->w_out2 || !d->k_ctx || !d->vt_ctx || !d->vec || !d->y)
    return upk_fail(ctx, UPK_EINVAL, "cross_block: null operand");
  if (!upk_cross_block_supported(ctx, d))
    return upk_fail(ctx, UPK_ESHAPE, "cross_block: shape outside the fused kernel's domain (c=%d d=%d heads=%d n_kv=%d)",
                    d->c, d->d, d->heads, d->n_kv);
  hipStream_t stream = (hipStream_t)stream_;
  XbCfg* cfg = xb_find(d);
  XbArgs s;
  memset(&s, 0, sizeof(s));
  s.a1 = (const f16*)d->a1, s.t0 = (const f16*)d->t0, s.w1 = (const f16*)d->w_out1, s.wq = (const f16*)d->w_q;
  s.w3 = (const f16*)d->w_out2, s.kc = (const f16*)d->k_ctx, s.vt = (const f16*)d->vt_ctx;
  s.zero = (const f16*)ctx->zero_page, s.vec = d->vec, s.y = (f16*)d->y;
  s.lda = d->lda, s.ldt0 = d->ld_t0, s.ldy = d->ldy, s.ldk = d->ldk, s.vt_ld = d->vt_ld, s.M = d->m, s.hw = d->hw;
  s.nkv = d->n_kv;
  const int hd = d->heads * d->d;
  const int vec_len = 2 * d->c + 2 * hd;
  s.vec_pieces = (vec_len + 255) / 256;
  s.ln_inv_dim = 1.0f / (float)(d->ln_dim > 0 ? d->ln_dim : d->c);
  s.ln_eps = d->ln_eps;
  s.scale_log2 = d->scale * 1.44269504088896340736f;
#ifdef UPK_TIMELINE
  s.dbg = getenv("UPK_XB_TL") ? (unsigned long long*)((char*)ctx->ws + ctx->ws_bytes - 4096) : nullptr;
#endif
  const int bm = cfg->bm;
  const size_t lds = (size_t)(hd / 32 + d->c / 32) * bm * 64 + (size_t)bm * 8 + (size_t)s.vec_pieces * 1024;
  if (int rc = upk_lds_attr_once(ctx, (const void*)cfg->fn, &cfg->attr_done)) return rc;
  upk_prof_scope prof(ctx, UPK_CLS_ATTN, stream);
  hipLaunchKernelGGL(cfg->fn, dim3(d->m / bm), dim3(512), lds, stream, s);
  return upk_check_launch(ctx, "cross_block");
}

extern "C" int upk_head_block_supported(upk_ctx* ctx, const upk_hblock_desc* d) {
  if (!ctx || !d) return 0;
  if (d->heads != XB_NW || d->c != 224 || d->d != 32) return 0;
  const int bm = d->rows_per_wg > 0 ? d->rows_per_wg : 32;
  if (bm != 32 && bm != 16) return 0;
  if (d->hw <= 0 || d->hw % bm || d->m % d->hw || d->vt_ld < d->hw) return 0;
  if ((d->ldx & 7) || (d->ld_t0 & 3) || (d->ld_qk & 3)) return 0;
  return 1;
}

extern "C" int upk_head_block_f16(upk_ctx* ctx, const upk_hblock_desc* d, upk_stream stream_) {
  if (!ctx || !d) return UPK_EINVAL;
  if (!d->x || !d->w_in || !d->w_qkv || !d->vec || !d->t0 || !d->qk || !d->vt)
    return upk_fail(ctx, UPK_EINVAL, "head_block: null operand");
  if (!upk_head_block_supported(ctx, d))
    return upk_fail(ctx, UPK_ESHAPE, "head_block: shape outside the fused kernel's domain (c=%d d=%d heads=%d)", d->c, d->d,
                    d->heads);
  hipStream_t stream = (hipStream_t)stream_;
  HbArgs s;
  memset(&s, 0, sizeof(s));
  s.x = (const f16*)d->x, s.w1 = (const f16*)d->w_in, s.w2 = (const f16*)d->w_qkv, s.zero = (const f16*)ctx->zero_page;
  s.vec = d->vec, s.t0 = (f16*)d->t0, s.qk = (f16*)d->qk, s.vt = (f16*)d->vt;
  s.ldx = d->ldx, s.ldt0 = d->ld_t0, s.ldqk = d->ld_qk, s.vt_ld = d->vt_ld, s.M = d->m, s.hw = d->hw;
  const int hd = d->heads * d->d;
  s.vec_pieces = (d->c + 6 * hd + 255) / 256;
  s.ln_inv_dim = 1.0f / (float)(d->ln_dim > 0 ? d->ln_dim : d->c);
  s.ln_eps = d->ln_eps;
#ifdef UPK_TIMELINE
  s.dbg = getenv("UPK_XB_TL") ? (unsigned long long*)((char*)ctx->ws + ctx->ws_bytes - 4096) : nullptr;
#endif
  const int bm = d->rows_per_wg > 0 ? d->rows_per_wg : 32;
  const bool gnf = d->gn_part != nullptr;
  if (gnf) {
    if (!d->gn_gamma || !d->gn_beta || d->gn_groups <= 0 || d->c % d->gn_groups || d->gn_groups > UPK_GN_GROUPS_MAX ||
        d->gn_nblk <= 0 || d->gn_nblk > UPK_GN_MAX_CHUNKS || d->gn_ld < d->c)
      return upk_fail(ctx, UPK_EINVAL, "head_block: bad GroupNorm operands (groups=%d nblk=%d ld=%d)", d->gn_groups,
                      d->gn_nblk, d->gn_ld);
    s.gn_part = d->gn_part, s.gn_gamma = d->gn_gamma, s.gn_beta = d->gn_beta;
    s.gn_nblk = d->gn_nblk, s.gn_ld = d->gn_ld, s.gn_cpg = d->c / d->gn_groups, s.gn_eps = d->gn_eps;
  }
  void (*fn)(const HbArgs) = bm == 32 ? (gnf ? hblock_kernel<2, 7, 32, true> : hblock_kernel<2, 7, 32, false>)
                                      : (gnf ? hblock_kernel<1, 7, 32, true> : hblock_kernel<1, 7, 32, false>);
  const size_t lds = (size_t)2 * (d->c / 32) * bm * 64 + (size_t)bm * 8 + (size_t)s.vec_pieces * 1024 +
                     (gnf ? (size_t)2 * d->c * 4 + 2 * UPK_GN_GROUPS_MAX * 8 + 2 * UPK_GN_GROUPS_MAX * 4 : 0);
  static unsigned long long hb_attr[4] = {};
  if (int rc = upk_lds_attr_once(ctx, (const void*)fn, &hb_attr[(bm == 32 ? 0 : 2) + (gnf ? 1 : 0)])) return rc;
  upk_prof_scope prof(ctx, UPK_CLS_IGEMM, stream);
  hipLaunchKernelGGL(fn, dim3(d->m / bm), dim3(512), lds, stream, s);
  return upk_check_launch(ctx, "head_block");
}

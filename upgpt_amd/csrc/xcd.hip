// Per-XCD persistent engine (include/upk.h upk_xcd_run_f16): a whole SpatialTransformer (attention.py:250-261 around the
// BasicTransformerBlock of :211-215) as ONE launch instead of 9-10.
//
// Why this shape (DESIGN.md 12a has the arithmetic and the measurements):
//   * every op of the block is per sample, and B = 8 samples meet 8 XCDs: sample b lives on XCD b % 8, its activations
//     (<= 0.5 MB per tensor) never leave that XCD's 4 MB L2;
//   * the 32 CUs of an XCD split each op — a GEMM as a pm x pn grid of (rows x output-column tiles), attention as
//     heads x query tiles, GroupNorm as one group per CU (32 groups = 32 CUs) — so a CU streams only its slice of a weight;
//   * a phase edge is an XCD-LOCAL barrier: plain stores, s_waitcnt vmcnt(0) (data is in the shared L2), one atomic on a
//     counter, sc1 (L1-bypassing) polls, sc1 loads of what other CUs wrote.  1.04 us measured, against 4.9 us for an
//     agent-scope release / acquire pair and ~2 us + cold first touches for a kernel boundary;
//   * the price is that every XCD streams every weight of the block through the fabric (7.4 TB/s for eight replicas of a
//     stream, scripts/ubench/xcdsync.hip): 72 MB per block at 16x16, 266 MB at 8x8 — which is why the weight-heavy
//     ResBlock convs (14-29 MB each) stay chip-wide launches.
//
// GEMM phase: the CU stages its mb (<= 64) rows of A ONCE in LDS (through registers: LayerNorm is applied on the way,
// rows padded to K rounded up to 128 so the K loop has no tail), the weights never touch LDS: a wave streams the 1 KiB
// operand fragments of its tiles global -> VGPR through a 4-deep register ring and walks K without a barrier.  Operands
// are swapped (weights = MFMA A operand) so a lane owns 4 consecutive output channels of one token; V tiles of the q|k|v
// GEMM swap them back, which transposes the accumulator: V^T leaves in 8-byte stores.
#include "common.h"

namespace {

static_assert(sizeof(upk_xphase) == 216, "upk_xphase layout differs from the ctypes mirror (upgpt_amd/_lib.py XPhase)");
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));

constexpr int XT = 512;              // threads per workgroup (8 waves, two per SIMD)
constexpr int X_LDS = 152 * 1024;    // dynamic LDS: one workgroup per CU by construction (the last 16 bytes: barrier flag)
constexpr int X_SC1 = 16;            // buffer aux bit: sc1 (bypass the CU's L1, served by the XCD's L2)
constexpr int X_WORDS = 64;          // u32 words per XCD in the sync workspace (arrive @0, barrier @16, exit @32)
constexpr int X_STATUS = 8 * X_WORDS;
constexpr u32 X_SPIN_LIMIT = 1u << 21;
constexpr int X_GN_MAXV = 16;        // values per thread a (sample, group) may take (n * C / groups <= 8192)
constexpr int X_RING = 4;            // weight fragments in flight per tile stream

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* p, u32 bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32 poll_sc1(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// XCD-local barrier: every wave's stores are in the L2 before the workgroup arrives; one lane polls.  Returns false
// (for the whole workgroup) when the XCD's other workgroups did not show up in time: `status` is set, the caller leaves.
__device__ __forceinline__ bool xbarrier(u32* cnt, u32 target, u32* status, int* flag_lds) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u32 spins = 0;
    int ok = 1;
    while (poll_sc1(cnt) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > X_SPIN_LIMIT || ((spins & 1023u) == 0 && poll_sc1(status) != 0)) {
        __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
    *flag_lds = ok;
  }
  __syncthreads();
  const int ok = __builtin_amdgcn_readfirstlane(*(volatile int*)flag_lds);
  __syncthreads();
  return ok != 0;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------ GroupNorm phase
// CU r owns group r of the sample: its n x cg values stay in registers between the statistics and the output (two exact
// passes, fp32).  Element e = tid + 512 i sits at (token e / cg, channel e % cg): walked incrementally, no divisions.
__device__ __forceinline__ void gn_phase(const upk_xphase& p, int b, int rank, char* lds) {
  float* red = (float*)lds;  // [2][8] wave partials
  const int cg = p.k1 / p.groups;
  const int cnt = p.n * cg;
  const f16* x = (const f16*)p.a + (size_t)b * p.n * p.lda;
  f16* y = (f16*)p.y + (size_t)b * p.n * p.ldy;
  const __amdgpu_buffer_rsrc_t rs = mk_rsrc(x, (u32)((size_t)p.n * p.lda * 2));
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tok0 = (int)threadIdx.x / cg, ch0 = (int)threadIdx.x - tok0 * cg;
  const int dq = XT / cg, dr = XT - dq * cg;  // e += 512  ->  (tok += dq, ch += dr) with one carry
  for (int grp = rank; grp < p.groups; grp += 32) {
    float v[X_GN_MAXV];
    float s = 0.f;
    int tok = tok0, ch = ch0;
#pragma unroll
    for (int i = 0; i < X_GN_MAXV; ++i) {
      const u32 off = tok < p.n ? (u32)((tok * p.lda + grp * cg + ch) * 2) : 0x80000000u;  // (out of range reads 0)
      const unsigned short h = __builtin_amdgcn_raw_buffer_load_b16(rs, off, 0, X_SC1);
      v[i] = (float)__builtin_bit_cast(f16, h);
      s += v[i];
      tok += dq;
      ch += dr;
      if (ch >= cg) {
        ch -= cg;
        ++tok;
      }
    }
    s = wave_sum(s);
    __syncthreads();
    if (lane == 0) red[wave] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) tot += red[w];
    const float mean = tot / (float)cnt;
    float ss = 0.f;
    tok = tok0, ch = ch0;
#pragma unroll
    for (int i = 0; i < X_GN_MAXV; ++i) {
      const float d = tok < p.n ? v[i] - mean : 0.f;
      ss += d * d;
      tok += dq;
      ch += dr;
      if (ch >= cg) {
        ch -= cg;
        ++tok;
      }
    }
    ss = wave_sum(ss);
    if (lane == 0) red[8 + wave] = ss;
    __syncthreads();
    float tot2 = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) tot2 += red[8 + w];
    const float rstd = rsqrtf(tot2 / (float)cnt + p.eps);
    tok = tok0, ch = ch0;
#pragma unroll
    for (int i = 0; i < X_GN_MAXV; ++i) {
      if (tok < p.n) {
        float o = (v[i] - mean) * rstd;
        if (p.gamma) o = o * p.gamma[grp * cg + ch] + p.beta[grp * cg + ch];
        if (p.silu) o = upk_silu(o);
        y[(size_t)tok * p.ldy + grp * cg + ch] = (f16)o;
      }
      tok += dq;
      ch += dr;
      if (ch >= cg) {
        ch -= cg;
        ++tok;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ GEMM phase
// Stages rows [r0, r0 + mb) of sample b of [a | a2] into LDS: row stride SA bytes (= Kpad * 2 + 32: fragment reads are
// conflict-free with a stride of 32 mod 64 bytes), columns [K, Kpad) and rows >= n zero.
__device__ __forceinline__ void stage_a(const upk_xphase& p, int b, int r0, int mb, char* lds, int SA, int Kpad) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = p.n, k1 = p.k1, K = p.k1 + p.k2;
  const f16* a1 = (const f16*)p.a + (size_t)b * n * p.lda;
  const __amdgpu_buffer_rsrc_t rs1 = mk_rsrc(a1, (u32)((size_t)n * p.lda * 2));
  if (p.ln) {
    // LayerNorm over the k1 (<= 1024) columns of a row on the way: (x - mean) * rstd, the affine lives in W / bias
    const float inv = 1.0f / (float)k1;
    for (int row = wave; row < mb; row += 8) {
      const int tok = r0 + row;
      f16x8 v[2];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int k = (lane + 64 * i) * 8;
        const u32 off = (tok < n && k < k1) ? (u32)((tok * p.lda + k) * 2) : 0x80000000u;
        v[i] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, X_SC1));
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)v[i][e];
      }
      const float mean = wave_sum(s) * inv;
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int k = (lane + 64 * i) * 8;
        if (k < k1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = (float)v[i][e] - mean;
            ss += d * d;
          }
        }
      }
      const float rstd = rsqrtf(wave_sum(ss) * inv + p.eps);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int k = (lane + 64 * i) * 8;
        if (k < Kpad) {
          f16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (tok < n && k < k1) ? (f16)(((float)v[i][e] - mean) * rstd) : (f16)0.f;
          *(f16x8*)(lds + row * SA + k * 2) = o;
        }
      }
    }
  } else {
    const f16* a2 = (const f16*)p.a2 + (p.a2 ? (size_t)b * n * p.lda2 : 0);
    const __amdgpu_buffer_rsrc_t rs2 = mk_rsrc(p.a2 ? a2 : a1, p.a2 ? (u32)((size_t)n * p.lda2 * 2) : 0u);
    const int pieces = Kpad >> 3;
    for (int row = wave; row < mb; row += 8) {
      const int tok = r0 + row;
      for (int pc = lane; pc < pieces; pc += 64) {
        const int k = pc * 8;
        const u32 o1 = (tok < n && k < k1) ? (u32)((tok * p.lda + k) * 2) : 0x80000000u;
        const u32 o2 = (tok < n && k >= k1 && k < K) ? (u32)((tok * p.lda2 + (k - k1)) * 2) : 0x80000000u;
        const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs1, o1, 0, X_SC1);
        const u32x4 v2 = __builtin_amdgcn_raw_buffer_load_b128(rs2, o2, 0, X_SC1);
        *(u32x4*)(lds + row * SA + k * 2) = v1 | v2;  // (the one out of range is zero)
      }
    }
  }
}

// One wave, TN column tiles x tm (<= 4) row tiles, whole K.  MODE 0: plain epilogue (bias, residual), 1: GEGLU over the
// (value, gate) tile pair, 2: V tiles of the q | k | v GEMM (operands swapped back: accumulator = [token][channel]).
template <int TN, int MODE>
__device__ __forceinline__ void gemm_unit(const upk_xphase& p, const char* lds, int SA, int KCpad, int tm, int b, int r0,
                                          int tile, int nvalid) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int KC = (p.k1 + p.k2) >> 5;
  const u32x4* wp[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) wp[t] = (const u32x4*)p.w + ((size_t)(tile + (t < nvalid ? t : 0)) * KC) * 64 + lane;
  f32x4 acc[4][TN];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 ring[X_RING][TN];
#pragma unroll
  for (int d = 0; d < X_RING; ++d)
#pragma unroll
    for (int t = 0; t < TN; ++t) ring[d][t] = wp[t][(size_t)(d < KC ? d : KC - 1) * 64];
  const char* abase = lds + j * SA + g * 16;
  for (int kc = 0; kc < KCpad; kc += X_RING) {
#pragma unroll
    for (int d = 0; d < X_RING; ++d) {
      f16x8 wf[TN];
#pragma unroll
      for (int t = 0; t < TN; ++t) wf[t] = __builtin_bit_cast(f16x8, ring[d][t]);
      const int nk = kc + d + X_RING;
#pragma unroll
      for (int t = 0; t < TN; ++t) ring[d][t] = wp[t][(size_t)(nk < KC ? nk : KC - 1) * 64];  // (clamped: no branch around a load)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m < tm) {
          const f16x8 af = *(const f16x8*)(abase + m * 16 * SA + (kc + d) * 64);  // (zero beyond K: the padded chunks add 0)
#pragma unroll
          for (int t = 0; t < TN; ++t)
            acc[m][t] = MODE == 2 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(af, wf[t], acc[m][t], 0, 0, 0)
                                  : __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t], af, acc[m][t], 0, 0, 0);
        }
      }
    }
  }
  // ---- epilogue
  const int n = p.n;
  if (MODE == 2) {
    // acc[m][t][r] = V[token 16 m + 4 g + r][channel 16 tile + j]  ->  vt[(b, head, d)][token .. token + 3]
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      if (t < nvalid) {
        const int ch = (tile + t - p.vtile0) * 16 + j;
        const int head = ch / p.dp, dd = ch - head * p.dp;
        f16* dst = (f16*)p.vt + ((size_t)(b * p.heads + head) * p.dp + dd) * p.vt_ld;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int tok = r0 + 16 * m + 4 * g;
          if (m < tm && tok < n && head < p.heads) {
            const f16x4 o = {(f16)acc[m][t][0], (f16)acc[m][t][1], (f16)acc[m][t][2], (f16)acc[m][t][3]};
            *(f16x4*)(dst + tok) = o;
          }
        }
      }
    }
    return;
  }
  const __amdgpu_buffer_rsrc_t rr = mk_rsrc(p.res ? (const f16*)p.res + (size_t)b * n * p.ldres : (const f16*)p.y,
                                            p.res ? (u32)((size_t)n * p.ldres * 2) : 0u);
  f16* yb = (f16*)p.y + (size_t)b * n * p.ldy;
  if (MODE == 1) {
    const int ch0 = (tile >> 1) * 16 + 4 * g;
    const f32x4 bv = p.bias ? *(const f32x4*)(p.bias + tile * 16 + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 bg = p.bias ? *(const f32x4*)(p.bias + (tile + 1) * 16 + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int tok = r0 + 16 * m + j;
      if (m < tm && tok < n && ch0 < p.n_out) {
        f16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (f16)upk_geglu_mul(acc[m][0][r] + bv[r], acc[m][TN - 1][r] + bg[r]);
        *(f16x4*)(yb + (size_t)tok * p.ldy + ch0) = o;
      }
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    if (t < nvalid) {
      const int ch0 = (tile + t) * 16 + 4 * g;
      const f32x4 bb = p.bias ? *(const f32x4*)(p.bias + ch0) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int tok = r0 + 16 * m + j;
        if (m < tm && tok < n && ch0 < p.n_out) {
          f32x4 v = acc[m][t] + bb;
          if (p.res) {
            const u32x2 rv = __builtin_amdgcn_raw_buffer_load_b64(rr, (u32)((tok * p.ldres + ch0) * 2), 0, X_SC1);
            const f16x4 rh = __builtin_bit_cast(f16x4, rv);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)rh[r];
          }
          const f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
          *(f16x4*)(yb + (size_t)tok * p.ldy + ch0) = o;
        }
      }
    }
  }
}

__device__ __forceinline__ void gemm_phase(const upk_xphase& p, int b, int rank, char* lds) {
  if (rank >= p.pm * p.pn) return;
  const int im = rank / p.pn, in = rank - im * p.pn;
  const int r0 = im * p.mb;
  if (r0 >= p.n) return;
  const int K = p.k1 + p.k2;
  const int Kpad = (K + 127) & ~127;
  const int SA = Kpad * 2 + 32;
  stage_a(p, b, r0, p.mb, lds, SA, Kpad);
  __syncthreads();
  const int rows = p.n - r0 < p.mb ? p.n - r0 : p.mb;
  const int tm = (rows + 15) >> 4;
  const int tpc = (p.ntiles + p.pn - 1) / p.pn;
  const int t0 = in * tpc;
  const int t1 = t0 + tpc < p.ntiles ? t0 + tpc : p.ntiles;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int KCpad = Kpad >> 5;
  if (p.tn == 2) {
    for (int tile = t0 + 2 * wave; tile < t1; tile += 16) {
      const int nv = t1 - tile >= 2 ? 2 : 1;
      if (p.epi == UPK_XE_GEGLU)
        gemm_unit<2, 1>(p, lds, SA, KCpad, tm, b, r0, tile, nv);
      else if (p.epi == UPK_XE_QKV && tile >= p.vtile0)
        gemm_unit<2, 2>(p, lds, SA, KCpad, tm, b, r0, tile, nv);
      else
        gemm_unit<2, 0>(p, lds, SA, KCpad, tm, b, r0, tile, nv);
    }
  } else {
    for (int tile = t0 + wave; tile < t1; tile += 8) {
      if (p.epi == UPK_XE_QKV && tile >= p.vtile0)
        gemm_unit<1, 2>(p, lds, SA, KCpad, tm, b, r0, tile, 1);
      else
        gemm_unit<1, 0>(p, lds, SA, KCpad, tm, b, r0, tile, 1);
    }
  }
}

// ------------------------------------------------------------------------------------------------ attention phase
__device__ __forceinline__ float max4(float x) {  // over the lanes (c, c + 16, c + 32, c + 48): the key groups of a query
  unsigned u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u = __float_as_uint(fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])));
  const auto bq = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return fmaxf(__uint_as_float(bq[0]), __uint_as_float(bq[1]));
}

// One wave = 16 queries of one (sample, head); keys in chunks of 32; S^T tiles [key][query] so that the probabilities a
// lane holds ARE its operand of the second matmul under the key permutation key(8 g + 4 t + r) = 16 t + 4 g + r
// (attention.hip has the derivation); V^T read with the same permutation.
template <int DP>
__device__ __forceinline__ void attn_tile16(const upk_xphase& p, int b, int head, int qt) {
  constexpr int KD = DP / 32, DT = DP / 16;
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int n = p.n, nkv = p.nkv;
  const f16* qb = (const f16*)p.a + (size_t)b * n * p.lda;
  const f16* kb_ = (const f16*)p.kk + (size_t)b * p.kbs;
  const f16* vb = (const f16*)p.vv + (size_t)b * p.vbs + (size_t)head * DP * p.vt_ld;
  const __amdgpu_buffer_rsrc_t rq = mk_rsrc(qb, (u32)((size_t)n * p.lda * 2));
  const __amdgpu_buffer_rsrc_t rk = mk_rsrc(kb_, (u32)((size_t)nkv * p.ldk * 2));
  const __amdgpu_buffer_rsrc_t rv = mk_rsrc(vb, (u32)((size_t)DP * p.vt_ld * 2));
  const int q = qt * 16 + c;
  f16x8 qf[KD];
#pragma unroll
  for (int kd = 0; kd < KD; ++kd) {
    const u32 off = q < n ? (u32)((q * p.lda + head * DP + kd * 32 + 8 * g) * 2) : 0x80000000u;
    qf[kd] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, off, 0, X_SC1));
  }
  f32x4 o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 lsum = {0.f, 0.f, 0.f, 0.f};
  float mrun = -INFINITY;
  const float cs = p.scale_log2;
  const f16x8 ones = {(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};
  for (int kb = 0; kb < nkv; kb += 32) {
    f32x4 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int key = kb + 16 * t + c;
      const int kr = key < nkv ? key : nkv - 1;
#pragma unroll
      for (int kd = 0; kd < KD; ++kd) {
        const f16x8 kf = __builtin_bit_cast(
            f16x8, __builtin_amdgcn_raw_buffer_load_b128(rk, (u32)((kr * p.ldk + p.koff + head * DP + kd * 32 + 8 * g) * 2), 0, X_SC1));
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[kd], s[t], 0, 0, 0);
      }
    }
    // V^T fragments requested before the softmax arithmetic
    f16x4 va[DT], vc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const u32 off = (u32)(((16 * i + c) * p.vt_ld + kb + 4 * g) * 2);
      va[i] = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rv, off, 0, X_SC1));
      vc[i] = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rv, off + 32, 0, X_SC1));
    }
    if (kb + 32 > nkv) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kb + 16 * t + 4 * g + r >= nkv) s[t][r] = -INFINITY;
    }
    float mx = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])), fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
    mx = max4(mx);
    const float mnew = fmaxf(mrun, mx);  // (finite: every chunk has a visible key)
    if (__builtin_amdgcn_ballot_w64(mnew != mrun) != 0) {
      const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * cs);
      lsum *= alpha;
#pragma unroll
      for (int i = 0; i < DT; ++i) o[i] *= alpha;
    }
    mrun = mnew;
    const float mc = -mnew * cs;
    f16x8 pf;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) pf[t * 4 + r] = (f16)__builtin_amdgcn_exp2f(fmaf(s[t][r], cs, mc));
    lsum = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf, lsum, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const f16x8 vf = {va[i][0], va[i][1], va[i][2], va[i][3], vc[i][0], vc[i][1], vc[i][2], vc[i][3]};
      o[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[i], 0, 0, 0);
    }
  }
  if (q < n) {
    const float inv = 1.0f / lsum[0];
    f16* dst = (f16*)p.y + ((size_t)b * n + q) * p.ldy + head * DP + 4 * g;
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const f16x4 ov = {(f16)(o[i][0] * inv), (f16)(o[i][1] * inv), (f16)(o[i][2] * inv), (f16)(o[i][3] * inv)};
      *(f16x4*)(dst + 16 * i) = ov;
    }
  }
}

__device__ __forceinline__ void attn_phase(const upk_xphase& p, int b, int rank) {
  const int cph = 32 / p.heads;  // CUs per head
  const int head = rank / cph, part = rank - head * cph;
  if (head >= p.heads) return;
  const int nqt = (p.n + 15) >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int qt = part + cph * wave; qt < nqt; qt += cph * 8) {
    if (p.dp == 32)
      attn_tile16<32>(p, b, head, qt);
    else if (p.dp == 64)
      attn_tile16<64>(p, b, head, qt);
    else
      attn_tile16<128>(p, b, head, qt);
  }
}

// ------------------------------------------------------------------------------------------------ the engine
__global__ __launch_bounds__(XT) void xcd_engine_kernel(const upk_xphase* __restrict__ phases, int nphases, int batch,
                                                        u32* sync) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int xcc = xcc_id() & 7;
  u32* mine = sync + xcc * X_WORDS;
  u32* status = sync + X_STATUS;
  int* flag = (int*)(lds + X_LDS - 16);  // (LDS word of the barrier's verdict / the rank broadcast: behind every phase's use)
  if (threadIdx.x == 0) *flag = (int)__hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int rank = __builtin_amdgcn_readfirstlane(*(volatile int*)flag);
  __syncthreads();
  if (rank >= 32) {  // more than 32 workgroups on this XCD: a placement the protocol does not cover
    if (threadIdx.x == 0) __hip_atomic_store(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    // every XCD meets once before anything else: all 32 arrivals are in before the first workgroup can reach the exit
    u32 epoch = 1;
    bool ok = xbarrier(mine + 16, 32u, status, flag);
    for (int b = xcc; ok && b < batch; b += 8) {
      for (int ph = 0; ok && ph < nphases; ++ph) {
        const upk_xphase& p = phases[ph];
        if (p.kind == UPK_XP_GEMM)
          gemm_phase(p, b, rank, lds);
        else if (p.kind == UPK_XP_ATTN)
          attn_phase(p, b, rank);
        else
          gn_phase(p, b, rank, lds);
        ++epoch;
        ok = xbarrier(mine + 16, 32u * epoch, status, flag);
      }
    }
  }
  // leave the words zeroed for the next launch: the last workgroup of the XCD to get here resets its lines (nobody polls
  // them any more: every poller has passed its last barrier before it arrives at the exit counter)
  if (threadIdx.x == 0) {
    const u32 left = __hip_atomic_fetch_add(mine + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u32 here = poll_sc1(mine);
    if (left + 1 == here) {  // (arrivals are complete long before the first exit: a barrier of 32 lies between)
      if (here != 32u) __hip_atomic_store(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

extern "C" size_t upk_xcd_sync_bytes(void) { return (size_t)(8 * X_WORDS + 64) * 4; }

extern "C" int upk_xcd_phase_check(upk_ctx* ctx, const upk_xphase* p) {
  if (!ctx || !p) return UPK_EINVAL;
  if (p->n <= 0 || (p->n & 3)) return upk_fail(ctx, UPK_ESHAPE, "xcd phase: rows per sample (%d) must be a positive multiple of 4", p->n);
  if (p->kind == UPK_XP_GN) {
    if (!p->a || !p->y) return upk_fail(ctx, UPK_EINVAL, "xcd GroupNorm phase: null tensor");
    if (p->groups <= 0 || p->k1 % p->groups) return upk_fail(ctx, UPK_ESHAPE, "xcd GroupNorm phase: %d channels / %d groups", p->k1, p->groups);
    if ((long long)p->n * (p->k1 / p->groups) > (long long)X_GN_MAXV * XT)
      return upk_fail(ctx, UPK_ESHAPE, "xcd GroupNorm phase: a (sample, group) of %d x %d values does not fit the registers of a workgroup", p->n, p->k1 / p->groups);
    if ((!p->gamma) != (!p->beta)) return upk_fail(ctx, UPK_EINVAL, "xcd GroupNorm phase: gamma and beta come together");
    return UPK_OK;
  }
  if (p->kind == UPK_XP_GEMM) {
    if (!p->a || !p->w || !p->y) return upk_fail(ctx, UPK_EINVAL, "xcd GEMM phase: null tensor");
    const int K = p->k1 + p->k2;
    if (p->k1 <= 0 || (p->k1 & 31) || (p->k2 & 31) || (p->k2 > 0) != (p->a2 != nullptr))
      return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: k1 = %d, k2 = %d must be multiples of 32 (k2 with a2)", p->k1, p->k2);
    if (p->ln && (p->k2 || p->k1 > 1024)) return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: LayerNorm rows of %d (+%d) columns", p->k1, p->k2);
    if (p->pm <= 0 || p->pn <= 0 || p->pm * p->pn > 32) return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: CU grid %d x %d", p->pm, p->pn);
    if (p->mb <= 0 || p->mb > 64 || (p->mb & 15) || (long long)p->pm * p->mb < p->n)
      return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: %d row blocks of %d rows for %d rows", p->pm, p->mb, p->n);
    const int Kpad = (K + 127) & ~127;
    if ((long long)p->mb * (Kpad * 2 + 32) > X_LDS - 64) return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: %d rows x K = %d do not fit in LDS", p->mb, K);
    if (p->tn != 1 && p->tn != 2) return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: tn = %d", p->tn);
    if (p->epi == UPK_XE_GEGLU && (p->tn != 2 || (p->ntiles & 1) || ((p->ntiles + p->pn - 1) / p->pn & 1)))
      return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: GEGLU needs (value, gate) tile pairs per wave and per CU");
    if (p->epi == UPK_XE_QKV) {
      if (!p->vt || p->heads <= 0 || p->dp <= 0 || (p->dp & 15) || (p->vt_ld & 3))
        return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: q|k|v epilogue operands");
      const int tpc = (p->ntiles + p->pn - 1) / p->pn;
      if (p->tn == 2 && ((p->vtile0 & 1) || (tpc & 1))) return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: V tiles must start on a wave's tile pair");
    }
    if (p->n_out <= 0 || (p->n_out & 3) || p->ntiles <= 0) return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: n_out = %d", p->n_out);
    if ((p->lda & 7) || (p->a2 && (p->lda2 & 7)) || (p->ldy & 3) || (p->res && (p->ldres & 3)))
      return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: leading dimensions must keep 16-byte rows / 8-byte stores aligned");
    return UPK_OK;
  }
  if (p->kind == UPK_XP_ATTN) {
    if (!p->a || !p->kk || !p->vv || !p->y) return upk_fail(ctx, UPK_EINVAL, "xcd attention phase: null tensor");
    if (p->heads <= 0 || p->heads > 32 || (p->dp != 32 && p->dp != 64 && p->dp != 128))
      return upk_fail(ctx, UPK_ESHAPE, "xcd attention phase: heads = %d, head dim = %d", p->heads, p->dp);
    if (p->nkv <= 0 || p->vt_ld < ((p->nkv + 31) & ~31)) return upk_fail(ctx, UPK_ESHAPE, "xcd attention phase: %d keys, vt_ld = %d", p->nkv, p->vt_ld);
    if ((p->lda & 7) || (p->ldk & 7) || (p->koff & 7) || (p->vt_ld & 3) || (p->ldy & 3))
      return upk_fail(ctx, UPK_ESHAPE, "xcd attention phase: alignment of the operand rows");
    return UPK_OK;
  }
  return upk_fail(ctx, UPK_EINVAL, "xcd phase: unknown kind %d", p->kind);
}

extern "C" int upk_xcd_run_f16(upk_ctx* ctx, const upk_xphase* phases_dev, int nphases, int batch, void* sync_ws,
                               upk_stream stream) {
  if (!ctx || !phases_dev || !sync_ws || nphases <= 0 || batch <= 0) return upk_fail(ctx, UPK_EINVAL, "upk_xcd_run_f16: bad argument");
  if (ctx->num_cus != 256) return upk_fail(ctx, UPK_ESHAPE, "upk_xcd_run_f16: built for 8 XCDs x 32 CUs, device has %d CUs", ctx->num_cus);
  static unsigned long long attr_mask = 0;
  int rc = upk_lds_attr_once(ctx, (const void*)xcd_engine_kernel, &attr_mask);
  if (rc != UPK_OK) return rc;
  upk_prof_scope prof(ctx, UPK_CLS_IGEMM, (hipStream_t)stream);
  hipLaunchKernelGGL(xcd_engine_kernel, dim3(256), dim3(XT), X_LDS, (hipStream_t)stream, phases_dev, nphases, batch, (u32*)sync_ws);
  return upk_check_launch(ctx, "xcd_engine_kernel");
}

extern "C" int upk_xcd_status(upk_ctx* ctx, void* sync_ws, int* status_host) {
  if (!ctx || !sync_ws || !status_host) return UPK_EINVAL;
  UPK_HIP(ctx, hipDeviceSynchronize());
  u32 st = 0;
  UPK_HIP(ctx, hipMemcpy(&st, (const u32*)sync_ws + X_STATUS, 4, hipMemcpyDeviceToHost));
  *status_host = (int)st;
  if (st) UPK_HIP(ctx, hipMemset(sync_ws, 0, upk_xcd_sync_bytes()));
  return UPK_OK;
}

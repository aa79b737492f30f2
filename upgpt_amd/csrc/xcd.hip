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

static_assert(sizeof(upk_xphase) == 232, "upk_xphase layout differs from the ctypes mirror (upgpt_amd/_lib.py XPhase)");
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
// Pointers read out of a descriptor in memory have no address space the compiler can see: it would emit FLAT loads and
// stores for them (both wait counters, conservative vmcnt(0) everywhere: the first build of this file had 249 of them and
// a K loop five times slower than its bytes).  Everything a phase touches through such a pointer is global memory:
#define XG(T) __attribute__((address_space(1))) T
template <class T>
__device__ __forceinline__ XG(T)* as_global(T* p) {
  return (XG(T)*)p;
}
template <class T>
__device__ __forceinline__ const XG(T)* as_global(const T* p) {
  return (const XG(T)*)p;
}

constexpr int XT = 512;              // threads per workgroup (8 waves, two per SIMD)
constexpr int X_LDS = 152 * 1024;    // dynamic LDS: one workgroup per CU by construction (the last 16 bytes: barrier flag)
constexpr int X_SC1 = 16;            // buffer aux bit: sc1 (bypass the CU's L1, served by the XCD's L2)
constexpr int X_WORDS = 64;          // u32 words per XCD in the sync workspace (arrive @0, barrier @16, exit @32)
constexpr int X_STATUS = 8 * X_WORDS;
constexpr u32 X_SPIN_LIMIT = 1u << 21;
constexpr int X_GN_MAXV = 16;        // values per thread a (sample, group) may take (n * C / groups <= 8192)
constexpr int X_RING_FRAGS = 16;     // weight fragments (KiB) a wave keeps in flight: 16 / TN per tile stream

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* p, u32 bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32 poll_sc1(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// XCD-local barrier in two halves: every wave's stores are in the L2 before the workgroup arrives; between arrival and
// the wait a workgroup may start things that do not depend on the others (touch_next).  xb_wait returns false (for the
// whole workgroup) when the XCD's other workgroups did not show up in time: `status` is set, the caller leaves.
__device__ __forceinline__ void xb_arrive(u32* cnt) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool xb_wait(u32* cnt, u32 target, u32* status, int* flag_lds) {
  if (threadIdx.x == 0) {
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

// Sum over the 64 lanes, result in every lane: four DPP butterflies inside each row of 16 (VALU, no LDS crossbar round
// trips as __shfl_xor makes them), two permlane swaps across the rows.
__device__ __forceinline__ float dpp_add(float v, int ctrl_sel) {
  const int iv = __builtin_bit_cast(int, v);
  int o;
  if (ctrl_sel == 0)
    o = __builtin_amdgcn_update_dpp(0, iv, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
  else if (ctrl_sel == 1)
    o = __builtin_amdgcn_update_dpp(0, iv, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
  else if (ctrl_sel == 2)
    o = __builtin_amdgcn_update_dpp(0, iv, 0x141, 0xf, 0xf, true);  // row_half_mirror
  else
    o = __builtin_amdgcn_update_dpp(0, iv, 0x140, 0xf, 0xf, true);  // row_mirror
  return v + __builtin_bit_cast(float, o);
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add(v, 0);
  v = dpp_add(v, 1);
  v = dpp_add(v, 2);
  v = dpp_add(v, 3);
  unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u = __float_as_uint(__uint_as_float(a[0]) + __uint_as_float(a[1]));
  const auto c = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(c[0]) + __uint_as_float(c[1]);
}

// ------------------------------------------------------------------------------------------------ GroupNorm phase
// CU r owns group r of the sample: its n x cg values stay in registers between the statistics and the output (two exact
// passes, fp32).  Element e = tid + 512 i sits at (token e / cg, channel e % cg): walked incrementally, no divisions.
__device__ __forceinline__ void gn_phase(const upk_xphase& p, int b, int rank, char* lds) {
  float* red = (float*)lds;  // [2][8] wave partials
  const int cg = p.k1 / p.groups;
  const int cnt = p.n * cg;
  const f16* x = (const f16*)p.a + (size_t)b * p.n * p.lda;
  XG(f16)* y = as_global((f16*)p.y) + (size_t)b * p.n * p.ldy;
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
        if (p.gamma) o = o * as_global(p.gamma)[grp * cg + ch] + as_global(p.beta)[grp * cg + ch];
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
// Stages rows [r0, r0 + mb) of sample b of [a | a2] into LDS by LDS-DMA (global_load_lds_dwordx4 with sc1: the rows were
// written by other CUs of this XCD a phase ago): no VGPR round trip, no ds_write pass, every request of the tile in
// flight at once.  Row stride SA bytes = Kpad * 2 + 64 + 32: [K, Kpad) and one more chunk are zero (the K loop's tail and
// its out-of-slice chunks read them), + 32 makes the stride 32 mod 64 (conflict-free 16-byte fragment reads).  Rows past
// the sample's end re-read its last row: their outputs are never stored.
__device__ __forceinline__ void stage_a(const upk_xphase& p, int b, int r0, int mb, char* lds, int SA, int Kpad) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = p.n, k1 = p.k1, K = p.k1 + p.k2;
  const char* a1 = (const char*)p.a + (size_t)b * n * p.lda * 2;
  const char* a2 = p.a2 ? (const char*)p.a2 + (size_t)b * n * p.lda2 * 2 : a1;
  const int pieces = K >> 3;           // 16-byte pieces that carry data
  const int nseg = (pieces + 63) >> 6;  // 1 KiB DMA requests per row
  const int ptot = (Kpad + 32) >> 3;   // ... + zero tail + zero chunk
  for (int row = wave; row < mb; row += 8) {
    const int tok = r0 + row < n ? r0 + row : n - 1;
    for (int seg = 0; seg < nseg; ++seg) {
      const int pc = seg * 64 + lane, k = pc * 8;
      if (pc < pieces) {
        const char* src = k < k1 ? a1 + ((size_t)tok * p.lda + k) * 2 : a2 + ((size_t)tok * p.lda2 + (k - k1)) * 2;
        __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(lds + row * SA + seg * 1024), 16, 0, X_SC1);
      }
    }
    for (int pc = pieces + lane; pc < ptot; pc += 64) *(u32x4*)(lds + row * SA + pc * 16) = (u32x4){0u, 0u, 0u, 0u};
  }
}

// LayerNorm statistics of the staged rows (ln phases): wave w takes rows w, w + 8, ...; (mean, rstd) per row go to the
// table at the end of LDS.  The normalisation itself is algebra in the epilogue (the fold of igemm.hip's ln_colsum):
//   LN(x) W'^T = rstd (x W'^T - mean colsum(W')),  W' = W gamma, bias' = b + W beta.
__device__ __forceinline__ void ln_stats(const upk_xphase& p, int mb, const char* lds, int SA, float* stats) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k1 = p.k1;
  const float inv = 1.0f / (float)k1;
  const f16x2 one2 = {(f16)1.f, (f16)1.f};
  f16x8 v[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = wave + 8 * i, k = (lane + 64 * h) * 8;
      v[i][h] = (row < mb && k < k1) ? *(const f16x8*)(lds + row * SA + k * 2) : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f16x2 x2 = {v[i][h][2 * e], v[i][h][2 * e + 1]};
        s1 = __builtin_amdgcn_fdot2(x2, one2, s1, false);
        s2 = __builtin_amdgcn_fdot2(x2, x2, s2, false);
      }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const int row = wave + 8 * i;
    if (row < mb && lane == 0) {
      const float mean = s1 * inv;
      stats[2 * row] = mean;
      stats[2 * row + 1] = rsqrtf(fmaxf(s2 * inv - mean * mean, 0.f) + p.eps);
    }
  }
}

// One wave: TN column tiles x tm (<= 4) row tiles over the chunks [kc0, kc1) of K.  MODE 0: plain epilogue (bias,
// residual), 1: GEGLU over the (value, gate) tile pair, 2: V tiles of the q | k | v GEMM (operands swapped back:
// accumulator = [token][channel]).  wk > 1: the CU's waves form a (8 / wk) x wk grid, the wk waves of a tile group split K
// and meet in LDS (the A tile is dead by then); every wave of the workgroup walks the same barriers, `valid` or not.
template <int TN, int MODE, int TM>
__device__ __forceinline__ void gemm_unit(const upk_xphase& p, char* lds, int SA, int zoff, int b, int r0, int tile, int nvalid,
                                          bool valid, int kc0, int kc1, int wk, int wki, int wn, const float* stats, long long* stamp) {
  constexpr int tm = TM;
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int KC = (p.k1 + p.k2) >> 5;
  const XG(u32x4)* wp[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) wp[t] = as_global((const u32x4*)p.w) + ((size_t)(tile + (t < nvalid ? t : 0)) * KC) * 64 + lane;
  f32x4 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // the weight ring first: the vector-memory counter retires in order, so the K loop's first wait must not stand behind
  // the (HBM-cold) bias line — epilogue operands are requested right after it and land while the loop runs
  constexpr int X_RING = X_RING_FRAGS / TN;
  u32x4 ring[X_RING][TN];
#pragma unroll
  for (int d = 0; d < X_RING; ++d)
#pragma unroll
    for (int t = 0; t < TN; ++t) ring[d][t] = wp[t][(size_t)(kc0 + d < KC ? kc0 + d : KC - 1) * 64];
  const int n = p.n;
  const __amdgpu_buffer_rsrc_t rr = mk_rsrc(p.res ? (const f16*)p.res + (size_t)b * n * p.ldres : (const f16*)p.y,
                                            p.res ? (u32)((size_t)n * p.ldres * 2) : 0u);
  u32x2 rv[TM][TN];
  f32x4 bia[TN], csum[TN];
  const bool ln = p.ln != 0;
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    if (MODE == 2) {
      const float cj = (ln && t < nvalid && wki == 0) ? as_global(p.colsum)[(tile + t) * 16 + j] : 0.f;
      csum[t] = (f32x4){cj, cj, cj, cj};
    } else {
      csum[t] = (ln && t < nvalid && wki == 0) ? *(const XG(f32x4)*)as_global(p.colsum + (tile + t) * 16 + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  if (MODE == 0) {
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const int ch0 = (tile + t) * 16 + 4 * g;
      bia[t] = (p.bias && t < nvalid && wki == 0) ? *(const XG(f32x4)*)as_global(p.bias + ch0) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        const int tok = r0 + 16 * m + j;
        const bool live = p.res && wki == 0 && valid && t < nvalid && tok < n && ch0 < p.n_out;
        rv[m][t] = __builtin_amdgcn_raw_buffer_load_b64(rr, live ? (u32)((tok * p.ldres + ch0) * 2) : 0x80000000u, 0, X_SC1);
      }
    }
  } else if (MODE == 1) {
#pragma unroll
    for (int t = 0; t < TN; ++t)
      bia[t] = (p.bias && wki == 0) ? *(const XG(f32x4)*)as_global(p.bias + (tile + t) * 16 + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const float bj = (p.bias && t < nvalid && wki == 0) ? as_global(p.bias)[(tile + t) * 16 + j] : 0.f;  // (the folded LayerNorm's W beta term)
      bia[t] = (f32x4){bj, bj, bj, bj};
    }
  }
  // K loop without a branch: X_RING chunks per trip, every A fragment of a group of four chunks is requested from LDS
  // before the group's MFMAs (with a test per chunk each MFMA stood behind its own LDS round trip: ~300 clk apiece,
  // scripts/gpu_xcd_abl.sh); chunks past the wave's slice read the row's zero chunk (zoff) and add nothing
  constexpr int GRP = TM * TN >= 6 ? 2 : 4;  // chunks whose A fragments are read together (register budget)
  const char* abase = lds + j * SA + g * 16;
  for (int kc = kc0; kc < kc1; kc += X_RING) {
#pragma unroll
    for (int d4 = 0; d4 < X_RING; d4 += GRP) {
      f16x8 af[GRP][TM];
#pragma unroll
      for (int e = 0; e < GRP; ++e) {
        const int c = kc + d4 + e;
        const int coff = c < kc1 ? c * 64 : zoff;
#pragma unroll
        for (int m = 0; m < TM; ++m) af[e][m] = *(const f16x8*)(abase + m * 16 * SA + coff);
      }
#pragma unroll
      for (int e = 0; e < GRP; ++e) {
        const int d = d4 + e;
        f16x8 wf[TN];
#pragma unroll
        for (int t = 0; t < TN; ++t) wf[t] = __builtin_bit_cast(f16x8, ring[d][t]);
        const int nk = kc + d + X_RING;
#pragma unroll
        for (int t = 0; t < TN; ++t) ring[d][t] = wp[t][(size_t)(nk < KC ? nk : KC - 1) * 64];  // (clamped: no branch around a load)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int t = 0; t < TN; ++t)
            acc[m][t] = MODE == 2 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(af[e][m], wf[t], acc[m][t], 0, 0, 0)
                                  : __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t], af[e][m], acc[m][t], 0, 0, 0);
      }
    }
  }
  if (stamp) stamp[4] = __builtin_readcyclecounter() + (__builtin_bit_cast(int, acc[0][0][0]) & 1);
  if (wk > 1) {
    __syncthreads();  // every wave is done with the A tile: its space carries the partial sums
    float* red = (float*)lds + ((size_t)(wn * (wk - 1) + (wki - 1)) * TM * TN) * 256 + lane * 4;
    if (wki > 0) {
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int t = 0; t < TN; ++t)
          *(f32x4*)(red + (m * TN + t) * 256) = acc[m][t];
    }
    __syncthreads();
    if (wki > 0) return;
    for (int o = 1; o < wk; ++o) {
      const float* src = (const float*)lds + ((size_t)(wn * (wk - 1) + (o - 1)) * TM * TN) * 256 + lane * 4;
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int t = 0; t < TN; ++t)
          acc[m][t] += *(const f32x4*)(src + (m * TN + t) * 256);
    }
  }
  if (!valid) return;
  // ---- epilogue
  if (MODE == 2) {
    // acc[m][t][r] = V[token 16 m + 4 g + r][channel 16 tile + j]  ->  vt[(b, head, d)][token .. token + 3]
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      if (t < nvalid) {
        const int ch = (tile + t - p.vtile0) * 16 + j;
        const int head = ch / p.dp, dd = ch - head * p.dp;
        XG(f16)* dst = as_global((f16*)p.vt) + ((size_t)(b * p.heads + head) * p.dp + dd) * p.vt_ld;
        const float bj = bia[t][0];
#pragma unroll
        for (int m = 0; m < TM; ++m) {
          const int tok = r0 + 16 * m + 4 * g;
          if (tok < n && head < p.heads) {
            f32x4 v = acc[m][t];
            if (ln) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float mean = stats[2 * (16 * m + 4 * g + r)], rstd = stats[2 * (16 * m + 4 * g + r) + 1];
                v[r] = rstd * (v[r] - mean * csum[t][0]);
              }
            }
            const f16x4 o = {(f16)(v[0] + bj), (f16)(v[1] + bj), (f16)(v[2] + bj), (f16)(v[3] + bj)};
            *(XG(f16x4)*)(dst + tok) = o;
          }
        }
      }
    }
    return;
  }
  XG(f16)* yb = as_global((f16*)p.y) + (size_t)b * n * p.ldy;
  if (MODE == 1) {
    const int ch0 = (tile >> 1) * 16 + 4 * g;
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      const int tok = r0 + 16 * m + j;
      if (tok < n && ch0 < p.n_out) {
        f16x4 o;
        const float mean = ln ? stats[2 * (16 * m + j)] : 0.f, rstd = ln ? stats[2 * (16 * m + j) + 1] : 1.f;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          o[r] = (f16)upk_geglu_mul(rstd * (acc[m][0][r] - mean * csum[0][r]) + bia[0][r],
                                    rstd * (acc[m][TN - 1][r] - mean * csum[TN - 1][r]) + bia[TN - 1][r]);
        *(XG(f16x4)*)(yb + (size_t)tok * p.ldy + ch0) = o;
      }
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    if (t < nvalid) {
      const int ch0 = (tile + t) * 16 + 4 * g;
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        const int tok = r0 + 16 * m + j;
        if (tok < n && ch0 < p.n_out) {
          f32x4 v = acc[m][t];
          if (ln) {
            const float mean = stats[2 * (16 * m + j)], rstd = stats[2 * (16 * m + j) + 1];
            v = rstd * (v - mean * csum[t]);
          }
          v += bia[t];
          const f16x4 rh = __builtin_bit_cast(f16x4, rv[m][t]);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)rh[r];
          const f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
          *(XG(f16x4)*)(yb + (size_t)tok * p.ldy + ch0) = o;
        }
      }
    }
  }
}

// geometry of a GEMM phase on this CU (shared by the phase itself and by the prefetch of the phase before it)
struct XGeom {
  int live, r0, tm, t0, t1, Kpad, SA, kc_per, KCp, wn_n;
};
__device__ __forceinline__ XGeom gemm_geom(const upk_xphase& p, int rank) {
  XGeom q;
  const int im = rank / p.pn, in = rank - im * p.pn;
  q.r0 = im * p.mb;
  q.live = rank < p.pm * p.pn && q.r0 < p.n;
  const int K = p.k1 + p.k2;
  const int wk = p.wk > 1 ? p.wk : 1;
  const int KC = K >> 5;
  q.kc_per = (((KC + wk - 1) / wk) + 3) & ~3;  // chunks per K slice: a multiple of 4 (the rows are zero padded that far)
  q.KCp = q.kc_per * wk;
  q.Kpad = q.KCp * 32;
  q.SA = q.Kpad * 2 + 64 + 32;  // (+ one zero chunk per row for the K loop's tail, + 32: conflict-free fragment reads)
  const int rows = p.n - q.r0 < p.mb ? p.n - q.r0 : p.mb;
  q.tm = (rows + 15) >> 4;
  const int tpc = (p.ntiles + p.pn - 1) / p.pn;
  q.t0 = in * tpc;
  q.t1 = q.t0 + tpc < p.ntiles ? q.t0 + tpc : p.ntiles;
  q.wn_n = 8 / wk;
  return q;
}

__device__ __forceinline__ void gemm_phase(const upk_xphase& p, int b, int rank, char* lds, long long* stamp) {  // stamp: thread 0 of a timed run only
  const XGeom q = gemm_geom(p, rank);
  if (!q.live) return;
  if (stamp) stamp[1] = __builtin_readcyclecounter() + (q.SA & 1);  // (geometry = descriptor fields are in)
  stage_a(p, b, q.r0, p.mb, lds, q.SA, q.Kpad);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the DMA requests of this wave have landed)
  if (stamp) stamp[2] = __builtin_readcyclecounter();
  __syncthreads();
  float* stats = (float*)(lds + X_LDS - 16 - 512);  // [64 rows][mean, rstd]
  if (p.ln) {
    ln_stats(p, p.mb, lds, q.SA, stats);
    __syncthreads();
  }
  if (stamp) stamp[3] = __builtin_readcyclecounter();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wk = p.wk > 1 ? p.wk : 1;
  const int wn = wave % q.wn_n, wki = wave / q.wn_n;
  const int kc0 = wki * q.kc_per, kc1 = kc0 + q.kc_per;
  const int tn = p.tn;
  // wk > 1: one pass, every wave takes part in the reduction barriers; wk == 1: a wave walks its tile groups alone
  const int zoff = q.Kpad * 2;
  for (int tile = q.t0 + tn * wn; wk > 1 ? tile == q.t0 + tn * wn : tile < q.t1; tile += tn * q.wn_n) {
    const bool valid = tile < q.t1;
    const int tl_ = valid ? tile : q.t0;
    const int nv = tn == 2 ? (q.t1 - tl_ >= 2 ? 2 : 1) : 1;
#define XCD_UNIT(TN_, MODE_)                                                                                            \
  do {                                                                                                                  \
    if (q.tm == 1)                                                                                                      \
      gemm_unit<TN_, MODE_, 1>(p, lds, q.SA, zoff, b, q.r0, tl_, nv, valid, kc0, kc1, wk, wki, wn, stats, stamp);       \
    else if (q.tm == 2)                                                                                                 \
      gemm_unit<TN_, MODE_, 2>(p, lds, q.SA, zoff, b, q.r0, tl_, nv, valid, kc0, kc1, wk, wki, wn, stats, stamp);       \
    else if (q.tm == 3)                                                                                                 \
      gemm_unit<TN_, MODE_, 3>(p, lds, q.SA, zoff, b, q.r0, tl_, nv, valid, kc0, kc1, wk, wki, wn, stats, stamp);       \
    else                                                                                                                \
      gemm_unit<TN_, MODE_, 4>(p, lds, q.SA, zoff, b, q.r0, tl_, nv, valid, kc0, kc1, wk, wki, wn, stats, stamp);       \
  } while (0)
    if (tn == 2) {
      if (p.epi == UPK_XE_GEGLU)
        XCD_UNIT(2, 1);
      else if (p.epi == UPK_XE_QKV && tl_ >= p.vtile0)
        XCD_UNIT(2, 2);
      else
        XCD_UNIT(2, 0);
    } else {
      if (p.epi == UPK_XE_QKV && tl_ >= p.vtile0)
        XCD_UNIT(1, 2);
      else
        XCD_UNIT(1, 0);
    }
#undef XCD_UNIT
  }
}

// At the START of a phase every wave asks for its share of the weight slice this CU will stream in the NEXT GEMM phase
// (weights do not depend on any barrier): one 4-byte load per 128-byte line, up to X_TOUCH_KB per CU (an XCD's 32 CUs
// together stay below its 4 MB L2).  The lines cross the fabric while the current phase runs, the next K loop then walks
// L2-resident weights (~0.3 us per ring fill instead of 1.5-2 us, scripts/xcd_timeline.py).  The values only have to
// stay alive until the phase ends.
constexpr int X_TOUCH_KB = 96;
constexpr int X_TOUCH_N = 2;  // wave loads (8 KiB each) per wave: 8 waves x 2 x 8 KiB = 128 KiB >= X_TOUCH_KB
struct XTouch {
  u32 v[X_TOUCH_N];
};
__device__ __forceinline__ XTouch touch_next(const upk_xphase* phases, int ph, int nphases, int rank) {
  XTouch t;
#pragma unroll
  for (int i = 0; i < X_TOUCH_N; ++i) t.v[i] = 0u;
  // the next GEMM phase (attention / GroupNorm phases in between stream no weights)
  // (host: nx = index of the next GEMM phase, or -1; before phase 0: the first GEMM phase of the list)
  const int nx = ph < 0 ? (phases[0].kind == UPK_XP_GEMM ? 0 : phases[0].nx) : phases[ph].nx;
  if (nx < 0 || nx >= nphases) return t;
  const upk_xphase& p = phases[nx];
  const XGeom q = gemm_geom(p, rank);
  if (!q.live) return t;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int KC = (p.k1 + p.k2) >> 5;
  const int ntl = q.t1 - q.t0;
  // per tile stream the first `per` KiB (all of it when the slice is small): what the K loops need first
  int per = (X_TOUCH_KB / ntl) & ~7;
  if (per < 8) per = 8;
  if (per > KC) per = (KC + 7) & ~7;
  const int lpt = per >> 3;  // wave loads per tile
  const int total = ntl * lpt;
  const char* base = (const char*)p.w + (size_t)q.t0 * KC * 1024;
  const size_t limit = (size_t)ntl * KC * 1024;
#pragma unroll
  for (int i = 0; i < X_TOUCH_N; ++i) {
    const int it = wave + 8 * i;
    if (it < total) {
      const int tl = it / lpt, part = it - tl * lpt;
      size_t off = ((size_t)tl * KC + part * 8) * 1024 + lane * 128;
      if (off >= limit) off = 0;
      t.v[i] = *(const XG(u32)*)as_global(base + off);
    }
  }
  return t;
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
// (attention.hip has the derivation); V^T read with the same permutation.  K [nkp][SK bytes] and V^T [DP][SV bytes] of
// the (sample, head) sit in LDS (attn_phase): the chunk loop never waits on global memory.
template <int DP>
__device__ __forceinline__ void attn_tile16(const upk_xphase& p, int b, int head, int qt, const char* ksm, const char* vsm,
                                            int SK, int SV) {
  constexpr int KD = DP / 32, DT = DP / 16;
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int n = p.n, nkv = p.nkv;
  const f16* qb = (const f16*)p.a + (size_t)b * n * p.lda;
  const __amdgpu_buffer_rsrc_t rq = mk_rsrc(qb, (u32)((size_t)n * p.lda * 2));
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
  const char* kl = ksm + c * SK + g * 16;
  const char* vl = vsm + c * SV + g * 8;
  for (int kb = 0; kb < nkv; kb += 32) {
    f32x4 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kd = 0; kd < KD; ++kd) {
        const f16x8 kf = *(const f16x8*)(kl + (kb + 16 * t) * SK + kd * 64);
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[kd], s[t], 0, 0, 0);
      }
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
      const f16x4 va = *(const f16x4*)(vl + i * 16 * SV + kb * 2);
      const f16x4 vc = *(const f16x4*)(vl + i * 16 * SV + kb * 2 + 32);
      const f16x8 vf = {va[0], va[1], va[2], va[3], vc[0], vc[1], vc[2], vc[3]};
      o[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[i], 0, 0, 0);
    }
  }
  if (q < n) {
    const float inv = 1.0f / lsum[0];
    XG(f16)* dst = as_global((f16*)p.y) + ((size_t)b * n + q) * p.ldy + head * DP + 4 * g;
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const f16x4 ov = {(f16)(o[i][0] * inv), (f16)(o[i][1] * inv), (f16)(o[i][2] * inv), (f16)(o[i][3] * inv)};
      *(XG(f16x4)*)(dst + 16 * i) = ov;
    }
  }
}

__device__ __forceinline__ void attn_phase(const upk_xphase& p, int b, int rank, char* lds) {
  const int cph = 32 / p.heads;  // CUs per head
  const int head = rank / cph, part = rank - head * cph;
  if (head >= p.heads) return;
  const int nqt = (p.n + 15) >> 4;
  if (part >= nqt) return;  // (no query tile for this CU: nothing to stage either)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int DP = p.dp, nkv = p.nkv;
  const int nkp = (nkv + 31) & ~31;
  // K rows: dp * 2 + 32 bytes (stride 32 mod 64: conflict-free 16-byte fragment reads); V^T rows: keys * 2 rounded up to
  // 256 + 16 bytes (the 32 lanes of an 8-byte read group fall on 32 different 8-byte slots)
  const int SK = DP * 2 + 32, SV = ((nkp * 2 + 255) & ~255) + 16;
  char* ksm = lds;
  char* vsm = lds + nkp * SK;
  {
    const f16* kg = (const f16*)p.kk + (size_t)b * p.kbs;
    const f16* vg = (const f16*)p.vv + (size_t)b * p.vbs + (size_t)head * DP * p.vt_ld;
    const __amdgpu_buffer_rsrc_t rk = mk_rsrc(kg, (u32)((size_t)nkv * p.ldk * 2));
    const __amdgpu_buffer_rsrc_t rv = mk_rsrc(vg, (u32)((size_t)DP * p.vt_ld * 2));
    const int kpr = DP >> 3;              // 16-byte pieces per K row
    const int nk_items = nkp * kpr;
    const int vpr = nkp >> 3;             // 16-byte pieces per V^T row (8 keys each)
    const int nv_items = DP * vpr;
    const int total = nk_items + nv_items;
    for (int it0 = threadIdx.x; it0 < total; it0 += XT * 8) {
      u32x4 v[8];
      int dst[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int it = it0 + u * XT;
        if (it < nk_items) {
          const int key = it / kpr, pc = it - key * kpr;
          v[u] = __builtin_amdgcn_raw_buffer_load_b128(rk, key < nkv ? (u32)((key * p.ldk + p.koff + head * DP + pc * 8) * 2) : 0x80000000u, 0, X_SC1);
          dst[u] = key * SK + pc * 16;
        } else if (it < total) {
          const int j2 = it - nk_items;
          const int d = j2 / vpr, pc = j2 - d * vpr;
          v[u] = __builtin_amdgcn_raw_buffer_load_b128(rv, pc * 8 < p.vt_ld ? (u32)((d * p.vt_ld + pc * 8) * 2) : 0x80000000u, 0, X_SC1);
          dst[u] = nkp * SK + d * SV + pc * 16;
        } else {
          v[u] = (u32x4){0u, 0u, 0u, 0u};
          dst[u] = -1;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (dst[u] >= 0) *(u32x4*)(lds + dst[u]) = v[u];
    }
  }
  __syncthreads();
  for (int qt = part + cph * wave; qt < nqt; qt += cph * 8) {
    if (DP == 32)
      attn_tile16<32>(p, b, head, qt, ksm, vsm, SK, SV);
    else if (DP == 64)
      attn_tile16<64>(p, b, head, qt, ksm, vsm, SK, SV);
    else
      attn_tile16<128>(p, b, head, qt, ksm, vsm, SK, SV);
  }
}

// ------------------------------------------------------------------------------------------------ the engine
// tl != nullptr (dev tool, scripts/xcd_timeline.py): thread 0 of every workgroup stamps the shader clock at the start of a
// phase, after its A tile is staged (GEMM), at the end of its body and behind the barrier: tl[((wg * nphases) + ph) * 4 ..].
#define XCD_STAMP(slot)                                                                                      \
  do {                                                                                                       \
    if (tl && threadIdx.x == 0) tl[((size_t)(xcc * 32 + rank) * nphases + ph) * 8 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)

__global__ __launch_bounds__(XT) void xcd_engine_kernel(const upk_xphase* __restrict__ phases, int nphases, int batch,
                                                        u32* sync, long long* tl) {
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
    u32 sink = 0u;
    // the phase descriptors (a few 128-byte lines) into this XCD's L2 before anybody's scalar loads want them
    if (threadIdx.x * 128 < nphases * (int)sizeof(upk_xphase)) sink ^= *(const XG(u32)*)as_global((const char*)phases + threadIdx.x * 128);
    xb_arrive(mine + 16);
    XTouch tw = touch_next(phases, -1, xcc < batch ? nphases : 0, rank);
    bool ok = xb_wait(mine + 16, 32u, status, flag);
#pragma unroll
    for (int i = 0; i < X_TOUCH_N; ++i) sink ^= tw.v[i];
    for (int b = xcc; ok && b < batch; b += 8) {
      for (int ph = 0; ok && ph < nphases; ++ph) {
        const upk_xphase& p = phases[ph];
        XCD_STAMP(0);
        tw = touch_next(phases, ph, nphases, rank);
        if (p.kind == UPK_XP_GEMM)
          gemm_phase(p, b, rank, lds, (tl && threadIdx.x == 0) ? tl + ((size_t)(xcc * 32 + rank) * nphases + ph) * 8 : nullptr);
        else if (p.kind == UPK_XP_ATTN)
          attn_phase(p, b, rank, lds);
        else
          gn_phase(p, b, rank, lds);
        XCD_STAMP(5);
        ++epoch;
#pragma unroll
        for (int i = 0; i < X_TOUCH_N; ++i) sink ^= tw.v[i];
        xb_arrive(mine + 16);
        ok = xb_wait(mine + 16, 32u * epoch, status, flag);
        XCD_STAMP(6);
      }
    }
    if (sink == 0x9e3779b9u && nphases < 0) sync[X_STATUS + 1] = sink;  // (never true: keeps the touches alive)
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

static void* g_xcd_timeline = nullptr;  // dev tool: device buffer of 256 * nphases * 4 clock stamps, or nullptr
extern "C" void upk_xcd_dev_timeline(void* buf) { g_xcd_timeline = buf; }

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
    if (p->ln && !p->colsum) return upk_fail(ctx, UPK_EINVAL, "xcd GEMM phase: a folded LayerNorm needs the column sums of its weight");
    if (p->ln && (p->k2 || p->k1 > 1024)) return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: LayerNorm rows of %d (+%d) columns", p->k1, p->k2);
    if (p->pm <= 0 || p->pn <= 0 || p->pm * p->pn > 32) return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: CU grid %d x %d", p->pm, p->pn);
    if (p->mb <= 0 || p->mb > 64 || (p->mb & 15) || (long long)p->pm * p->mb < p->n)
      return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: %d row blocks of %d rows for %d rows", p->pm, p->mb, p->n);
    const int wk_ = p->wk > 1 ? p->wk : 1;
    const int Kpad = (((((K >> 5) + wk_ - 1) / wk_) + 3) & ~3) * wk_ * 32;
    if ((long long)p->mb * (Kpad * 2 + 96) > X_LDS - 640) return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: %d rows x K = %d do not fit in LDS", p->mb, K);
    if (p->tn != 1 && p->tn != 2) return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: tn = %d", p->tn);
    const int wk = p->wk > 1 ? p->wk : 1;
    if (wk != 1 && wk != 2 && wk != 4 && wk != 8) return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: wk = %d", p->wk);
    if (wk > 1 && (p->ntiles + p->pn - 1) / p->pn > p->tn * (8 / wk))
      return upk_fail(ctx, UPK_ESHAPE, "xcd GEMM phase: with K split over %d waves a CU takes at most %d tiles", wk, p->tn * (8 / wk));
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
    {
      const int nkp = (p->nkv + 31) & ~31;
      if ((long long)nkp * (p->dp * 2 + 32) + (long long)p->dp * (((nkp * 2 + 255) & ~255) + 16) > X_LDS - 64)
        return upk_fail(ctx, UPK_ESHAPE, "xcd attention phase: K and V^T of a head (%d keys x %d) do not fit in LDS", p->nkv, p->dp);
    }
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
  hipLaunchKernelGGL(xcd_engine_kernel, dim3(256), dim3(XT), X_LDS, (hipStream_t)stream, phases_dev, nphases, batch, (u32*)sync_ws,
                     (long long*)g_xcd_timeline);
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

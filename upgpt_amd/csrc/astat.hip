// A-stationary GEMM family of upk_conv2d_nhwc_f16 (1x1 conv / Linear launches with K resident in LDS).
//
// Why another family.  The wave-specialised implicit-GEMM kernels (igemm.hip) stream BOTH operands through an LDS ring
// with one workgroup barrier per 32..128-deep K stage; on the UNet's Linears (K = 224..1120, N up to 7168) that means
//   * the activation tile is fetched once per N tile (14x for the GEGLU projection at 32x32),
//   * every stage is a rendezvous of 8 waves, so one late DMA stalls the whole workgroup,
//   * the epilogue (GEGLU: an erf per output) runs after the K loop with the matrix pipe idle.
// Here the BM x K activation tile is staged ONCE (LDS-DMA, XOR-swizzled source chunks, conflict-free ds_read_b128 as in
// igemm.hip), every wave owns its own output columns and pulls its weight fragments STRAIGHT from global memory / L2
// into registers — in the packed layout [K/32][n_pad][32] a 16-column x 32-deep fragment is one contiguous 1 KiB run, a
// perfect wave load — through a PF-chunk deep register ring.  After the one barrier behind the tile load nothing in the
// kernel synchronises: a wave walks (pass, chunk) = its column groups x K at its own pace, eight waves (two per SIMD)
// drift apart, and one wave's epilogue VALU work runs under its SIMD partner's MFMAs.  Weight fragments never touch the
// LDS, whose only traffic is MI fragment reads per MI*NI MFMAs per wave.
//
// Scope: ksize 1, stride 1, no upsample (IgemmArgs::linear), any of the two-source / appended-segment inputs (they are
// just more K chunks of the resident tile), no split-K, K / 32 a multiple of the ring depth (7 or 8: the 7*32 channel
// family and the power-of-two family), BM * K * 2 B + scratch <= 160 KiB.  Epilogues: everything igemm_common.h offers
// (bias, timestep row vector, residual, SiLU, GEGLU, transposed-V tail, fp32 / NCHW output, folded LayerNorm with the
// statistics taken from the resident tile or from the producer's row sums, GroupNorm channel partials, LayerNorm row
// sums of the output).
#include <type_traits>

#include "igemm_common.h"

namespace upkd {
namespace {

constexpr int AS_NW = 8;  // waves per workgroup (two per SIMD)

// Everything the kernel touches on its way to the first load and in the straight-line epilogues, in ONE compact block
// at the head of the kernarg segment and requested at kernel entry in one batch: read lazily out of the 600-byte
// IgemmArgs, the prologue was a chain of a dozen dependent scalar-cache misses (in-kernel stamps: 4.0k cycles from
// entry to the first DMA with warm caches, 11k inside the forward, where 850 MB of weights have gone through the
// caches since the last launch).
struct AsArgs {
  const f16* x1;
  const f16* x2;
  const f16* x3;
  const f16* x4;
  const f16* w;
  const f16* zero;
  const float* bias;
  const f16* res;
  f16* y;
  const float* ln_u;
  const float* lnr_in;
  int e1, e2, e3;  // channel offsets where the 2nd / 3rd / 4th source start (multiples of 32)
  int ld1, ld2, ld3, ld4;
  int M, npad, n_out, nch, flags, ldr, ldy;
  int ppw, npass;
  int tiles_m, tiles_n, xm_pm, xm_pn, xm_mi, xm_nj;
  int lnr_slots;
  float ln_inv_dim, ln_eps;
};
#define AS_PIN(v) asm volatile("" ::"s"(v))

__device__ __forceinline__ bool as_tile_map(const AsArgs& a, int& tm, int& tn) {
  if (a.xm_pm == 0) {
    const int tile = blockIdx.x;
    tn = tile / a.tiles_m;
    tm = tile - tn * a.tiles_m;
    return true;
  }
  const int w = blockIdx.x;
  const int xcd = w & 7, l = w >> 3;
  const int xi = xcd / a.xm_pn, xj = xcd - xi * a.xm_pn;
  const int ul = l / a.xm_mi, tl = l - ul * a.xm_mi;  // (see tile_map: M tiles fastest inside an XCD)
  tm = xi * a.xm_mi + tl;
  tn = xj * a.xm_nj + ul;
  return tm < a.tiles_m && tn < a.tiles_n;
}

// GEN: the instantiation that also carries igemm_common.h's general epilogue (SiLU, fp32 / NCHW output, V^T tail, row
// vector, GroupNorm partials, LayerNorm row sums).  Separate, because its live ranges would cost the two straight-line
// epilogues (plain, GEGLU: every large launch of the UNet) their registers.
template <int MI, int NI, int PF, bool GEN>
__global__ __launch_bounds__(512) void igemm_as_kernel(const AsArgs s, const IgemmArgs a) {
  constexpr int NW = AS_NW;
  constexpr int BM = MI * 16;
  constexpr int PW = NW * NI * 16;  // output columns per pass of the workgroup
  constexpr bool RA = MI * NI <= 8;  // A fragments read one K step ahead + residual prefetch (register budget: 2 waves per SIMD)
  extern __shared__ __attribute__((aligned(16))) f16 smem[];

  // (one batch of scalar loads, one wait)
  AS_PIN(s.x1); AS_PIN(s.x2); AS_PIN(s.x3); AS_PIN(s.x4); AS_PIN(s.w); AS_PIN(s.zero);
  AS_PIN(s.e1); AS_PIN(s.e2); AS_PIN(s.e3); AS_PIN(s.ld1); AS_PIN(s.ld2); AS_PIN(s.ld3); AS_PIN(s.ld4);
  AS_PIN(s.M); AS_PIN(s.npad); AS_PIN(s.nch); AS_PIN(s.flags); AS_PIN(s.ppw); AS_PIN(s.npass);
  AS_PIN(s.tiles_m); AS_PIN(s.tiles_n); AS_PIN(s.xm_pm); AS_PIN(s.xm_pn); AS_PIN(s.xm_mi); AS_PIN(s.xm_nj);
  AS_PIN(s.bias); AS_PIN(s.res); AS_PIN(s.y); AS_PIN(s.ln_u); AS_PIN(s.lnr_in); AS_PIN(s.n_out); AS_PIN(s.ldr);
  AS_PIN(s.ldy); AS_PIN(s.lnr_slots);
#undef ABL_ON
#if defined(UPK_DEV)  // (phase ablation hooks: dev builds only — as runtime tests they cost branches in every K step)
#define ABL_ON(f) ((s.flags & (f)) != 0)
#else
#define ABL_ON(f) (false)
#endif
  if ABL_ON(ABL_EMPTY) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4, lc = lane & 15;
#ifdef UPK_TIMELINE
  // dev: s_memtime stamps of waves 0 and 4 of the first and the last workgroup (scripts/timeline_as.py)
  const bool tl = (s.flags & ABL_TIMELINE) && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && (wave == 0 || wave == 4);
  unsigned long long* tlp = a.dbg + (blockIdx.x == 0 ? 0 : 64) + (wave == 0 ? 0 : 32);
#define STAMP(i) do { if (tl && lane == 0 && (i) < 32) tlp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
  STAMP(0);
  int tm, tn;
  if (!as_tile_map(s, tm, tn)) return;
  const int m0 = tm * BM;
  const int nch = s.nch;
  const int p0 = tn * s.ppw;
  const int p1 = min(s.npass, p0 + s.ppw);
  const bool geglu = s.flags & UPK_F_GEGLU;
  const bool geglu2 = (NI == 2) && geglu;

  // ---- 1. the activation tile: nch x MI row groups of 1 KiB, LDS-DMA, round-robin over the waves
  {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int r16 = lane >> 2;
    const int chd = (lane & 3) ^ ((-(lane >> 4)) & 3);  // source chunk of this lane (swizzle on the SOURCE side)
    const f16* zsrc = s.zero + (lane & 3) * 8;
    // one source at a time (a source chosen per chunk inside ONE loop becomes an indexed scratch array: a scratch load
    // and a vmcnt(0) — which also drains every DMA in flight — per 1 KiB request)
    auto issue = [&](const f16* sp, int sld, int c_lo, int c_hi) {
      const int lo = (c_lo >> 5) * MI, hi = (c_hi >> 5) * MI;
      for (int idx = lo + ((wave - lo) & (NW - 1)); idx < hi; idx += NW) {
        const int kc = idx / MI, rg = idx - kc * MI;
        const int m = m0 + rg * 16 + r16;
        const f16* src = m < s.M ? sp + (long)m * sld + (kc * 32 - c_lo) + chd * 8 : zsrc;
        f16* dst = smem + (kc * BM + rg * 16) * 32;
        __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)dst, 16, 0, 0);
      }
    };
    issue(s.x1, s.ld1, 0, s.e1);
    if (s.e2 > s.e1) issue(s.x2, s.ld2, s.e1, s.e2);
    if (s.e3 > s.e2) issue(s.x3, s.ld3, s.e2, s.e3);
    if (nch * 32 > s.e3) issue(s.x4, s.ld4, s.e3, nch * 32);
    // bias and LayerNorm column sums of this workgroup's columns [p0 * PW, p1 * PW), into LDS with the tile: read from
    // global memory per pass they put a vmcnt(0) in front of every epilogue (the compiler cannot count across the K
    // loop): a drain of the PF * NI weight fragments just requested for the next pass, ~2k cycles per pass
    if constexpr (!GEN) {
      float* bl0 = (float*)(smem + nch * BM * 32) + BM * 2 + NW * NI * 32;
      const int npc = (p1 - p0) * (PW / 256);  // 1 KiB pieces per array
      for (int idx = wave; idx < 2 * npc; idx += NW) {
        const bool second = idx >= npc;
        const int pc = second ? idx - npc : idx;
        const int col = p0 * PW + pc * 256 + lane * 4;
        const float* arr = second ? s.ln_u : s.bias;
        const float* src = (arr && col < s.npad) ? arr + col : (const float*)s.zero;
        __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(bl0 + (second ? s.ppw * PW : 0) + pc * 256), 16, 0, 0);
      }
    }
  }

  // ---- 2. this wave's output columns; weight addressing = wave-uniform base (SGPR pair, one per fragment and ring
  // block) + a per-lane, per-ring-slot 32-bit offset: nothing but the loads themselves in the K loop
  const int col0 = geglu2 ? (wave >> 1) * 64 + (wave & 1) * 16 : wave * NI * 16;  // packed column of fragment 0, pass 0
  const unsigned jstep = geglu2 ? 2048u : 1024u;                                  // bytes between a wave's fragments
  // passes [p0, p1w) have columns for this wave (the last pass of a launch may be ragged: npad is no multiple of PW)
  const int p1w = min(p1, (s.npad - col0 + PW - 1) / PW);
  const bool active = p0 < p1w;
  const char* wb = (const char*)s.w;
  const unsigned kstride = (unsigned)s.npad * 64u;  // bytes between K chunks of the packed weight
  // per-lane, per-ring-slot 32-bit offsets + one wave-uniform base (SGPR pair) per fragment and ring block: the K loop
  // holds the loads themselves and nothing else (a wave issues ~1 instruction per 4-5 cycles; a step of MI*NI MFMAs is
  // 128-256 matrix-pipe cycles: 20 scalar instructions of cursor arithmetic per step made the loop issue-bound)
  unsigned voff[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) voff[u] = (unsigned)(lc * 32 + lg * 8) * 2u + (unsigned)u * kstride;
  auto wbase = [&](int p, int kc) -> const char* {  // (p, kc) -> fragment 0's chunk, wave-uniform
    return wb + (size_t)(((unsigned)kc * (unsigned)s.npad + (unsigned)(col0 + p * PW)) * 64u);
  };
  STAMP(1);
  f16x8 ring[PF][NI];
  if (active) {
    const char* b0 = wbase(p0, 0);
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
      for (int j = 0; j < NI; ++j) ring[u][j] = *(const f16x8*)(b0 + j * jstep + voff[u]);
  }
  // which epilogue (workgroup-uniform): the two straight-line ones with their operands requested ahead of the pass's K
  // loop (a dependent load round trip behind the K loop costs ~1 us per pass), or igemm_common.h's general one
  const bool ep_geglu = !GEN && geglu;  // (the host picks GEN = false only for these two)
  const bool ep_plain = !GEN && !geglu;
  const f16* resp = s.res ? s.res : s.zero;
  const unsigned has_r = s.res ? ~0u : 0u;

  f16x4 rr[RA ? MI : 1][NI];
  auto epi_prefetch = [&](int p) {  // (plain epilogue: the residual rows of the pass, requested ahead of its K loop)
    const int nw = col0 + p * PW;
    if (ep_plain && RA) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const unsigned m = (unsigned)min(m0 + i * 16 + lc, s.M - 1);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const unsigned n = (unsigned)(nw + j * 16 + lg * 4);
          rr[i][j] = *(const f16x4*)(resp + ((m * (unsigned)s.ldr + (n < (unsigned)s.n_out ? n : 0u)) & has_r));
        }
      }
    }
  };
  if (active) epi_prefetch(p0);  // (with the ring, ahead of the barrier: one memory round trip for all of it)
  STAMP(2);
  __syncthreads();  // (drains the tile DMAs: vmcnt(0))
  STAMP(3);

  // ---- 3. folded LayerNorm: {mean, rstd} of the tile's rows, from the resident tile or from the producer's row sums
  float* st = (float*)(smem + nch * BM * 32);  // [BM][2]
  float* red = st + BM * 2;                    // Epi::tile_plain_cp scratch
  const float* bl = red + NW * NI * 32;        // [2][ppw * PW]: bias | LayerNorm column sums of columns p0 * PW ..
  if (s.ln_u) {
    if (s.lnr_in) {
      if (tid < BM) {  // (Epi::lnr_row: the producer's per-slot row sums)
        const int m = m0 + tid;
        const float* p = s.lnr_in + (long)(m < s.M ? m : 0) * 2;
        const long sstride = (long)s.M * 2;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const f32x2 v = q < s.lnr_slots ? *(const f32x2*)(p + q * sstride) : (f32x2){0.f, 0.f};
          s1 += v[0];
          s2 += v[1];
        }
        const float mu = s1 * s.ln_inv_dim;
        st[2 * tid] = mu;
        st[2 * tid + 1] = rsqrtf(fmaxf(s2 * s.ln_inv_dim - mu * mu, 0.f) + s.ln_eps);
      }
    } else {
      constexpr int LPR = 512 / BM;  // lanes per row
      const int row = tid / LPR, part = tid - row * LPR;
      float s1 = 0.f, s2 = 0.f;
      const f16x2 one2 = {(f16)1.f, (f16)1.f};
      for (int q = part; q < nch * 4; q += LPR) {
        const f16x8 v = *(const f16x8*)(smem + ((q >> 2) * BM + row) * 32 + (q & 3) * 8);
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
    __syncthreads();
  }

  // which epilogue (workgroup-uniform): the two straight-line ones with their operands requested at the START of the
  // pass (a dependent load round trip behind the K loop costs ~1 us per pass), or igemm_common.h's general one
  STAMP(4);
  // ---- 4. passes: K loop out of the resident tile + the register ring, then the epilogue of the pass
  const unsigned la = (unsigned)(lc * 32 + lds_swz(lc, lg) * 8) * 2u;  // this lane's byte offset inside a 16-row group
  const char* sm = (const char*)smem;
  f16x8 fa[MI];
  if (active && RA) {
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[i] = *(const f16x8*)(sm + la + i * 1024);
  }
  for (int p = p0; p < p1; ++p) {
    const int nw = col0 + p * PW;  // packed column of fragment 0 in this pass
    if (p > p0) epi_prefetch(p);   // (the first pass's were requested ahead of the tile barrier)
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p < p1w) {
#pragma unroll 1
      for (int kc0 = 0; kc0 < nch; kc0 += PF) {
        const bool wrap = kc0 + PF >= nch;       // the refills of this block belong to the next pass
        const bool dead = wrap && p + 1 >= p1w;  // ... which does not exist: every lane group re-reads the head of the
                                                 // weight (cache hits) instead of branching around the loads
        const char* sb[NI];
        {
          const char* b0 = dead ? wb : wbase(wrap ? p + 1 : p, wrap ? 0 : kc0 + PF);
#pragma unroll
          for (int j = 0; j < NI; ++j) sb[j] = b0 + (dead ? 0u : j * jstep);
        }
        const char* ldsA = sm + la + (unsigned)(kc0 * BM) * 64u;                    // chunks kc0 .. of the tile
        const char* ldsN = sm + la + (unsigned)((wrap ? 0 : kc0 + PF) * BM) * 64u;  // first chunk of the next block
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          f16x8 fc[MI];
          if constexpr (RA) {
#pragma unroll
            for (int i = 0; i < MI; ++i) fc[i] = fa[i];
            if (!ABL_ON(ABL_NOLDSW)) {
#pragma unroll
              for (int i = 0; i < MI; ++i)
                fa[i] = u + 1 < PF ? *(const f16x8*)(ldsA + ((u + 1) * BM + i * 16) * 64) : *(const f16x8*)(ldsN + i * 1024);
            }
          } else {
#pragma unroll
            for (int i = 0; i < MI; ++i) fc[i] = *(const f16x8*)(ldsA + (u * BM + i * 16) * 64);
          }
          if (!ABL_ON(ABL_NOMFMA)) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[u][j], fc[i], acc[i][j], 0, 0, 0);
          } else {
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(fc[i]));
#pragma unroll
            for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(ring[u][j]));
          }
          // refill the slot just consumed (same registers: the loads are issued behind the MFMAs that read them)
          if (!ABL_ON(ABL_NOGLOAD)) {
#pragma unroll
            for (int j = 0; j < NI; ++j) ring[u][j] = *(const f16x8*)(sb[j] + voff[u]);
          }
          __builtin_amdgcn_sched_barrier(0);  // (keeps the unrolled steps from hoisting each other's LDS reads)
        }
      }
    } else if (!GEN || !a.gn_cp) {
      break;  // (a wave past the last column; with channel partials it still joins the epilogue's barriers)
    }
    STAMP(5 + 2 * (p - p0));
    if ABL_ON(ABL_NOEPI) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(acc[i][j]));
      continue;
    }
    if constexpr (!GEN) {
      f32x4 bv[NI], lu[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int nl = nw - p0 * PW + (geglu2 ? j * 32 : j * 16) + lg * 4;
        bv[j] = *(const f32x4*)(bl + nl);
        lu[j] = *(const f32x4*)(bl + s.ppw * PW + nl);
      }
      if (s.ln_u) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const f32x2 mr = *(const f32x2*)(st + 2 * (i * 16 + lc));
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = (acc[i][j] - mr[0] * lu[j]) * mr[1];
        }
      }
      if (ep_geglu) {
        // packed columns: [32 value | 32 gate] per 64.  NI == 2: fragments (value, gate) 32 columns apart;
        // NI == 4: fragments 0, 1 are values, 2, 3 their gates
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int m = m0 + i * 16 + lc;
          f16* yrow = s.y + (unsigned)min(m, s.M - 1) * (unsigned)s.ldy;
#pragma unroll
          for (int jv = 0; jv < NI / 2; ++jv) {
            const int jg = jv + NI / 2;
            const int n = nw + jv * 16 + lg * 4;
            const int oc = (n >> 6) * 32 + (n & 31);
            const f32x4 v = acc[i][jv] + bv[jv], g = acc[i][jg] + bv[jg];
            f16x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (f16)upk_geglu_mul(v[k], g[k]);
            if (m < s.M && n < s.npad && oc < s.n_out) *(f16x4*)(yrow + oc) = o;
          }
        }
        STAMP(6 + 2 * (p - p0));
      } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int m = m0 + i * 16 + lc;
          const unsigned mm = (unsigned)min(m, s.M - 1);
          f16* yrow = s.y + mm * (unsigned)s.ldy;
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const int n = nw + j * 16 + lg * 4;
            f16x4 r;
            if constexpr (RA) r = rr[i][j];
            else r = *(const f16x4*)(resp + ((mm * (unsigned)s.ldr + (unsigned)(n < s.n_out ? n : 0)) & has_r));
            const f32x4 v = acc[i][j] + bv[j];
            f16x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (f16)(v[k] + (float)r[k]);
            if (m < s.M && n < s.n_out) *(f16x4*)(yrow + n) = o;
          }
        }
      }
    } else {
    if (a.ln_u) {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = nw + j * 16 + lg * 4;
        const f32x4 u = n < a.npad ? *(const f32x4*)(a.ln_u + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const f32x2 mr = *(const f32x2*)(st + 2 * (i * 16 + lc));
          acc[i][j] = (acc[i][j] - mr[0] * u) * mr[1];
        }
      }
    }
    Epi::tile<MI, NI, 1, NW>(a, m0, m0, nw, lc, lg, acc, 0, wave, red, a.M);
    }
  }
#ifdef UPK_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(31);
#endif
#undef STAMP
}

struct AsCfg {
  int mi, ni, pf;
  const char* name;
  void (*fn)(const AsArgs, const IgemmArgs);      // straight-line epilogues (plain / GEGLU)
  void (*fn_gen)(const AsArgs, const IgemmArgs);  // general epilogue
};
#define ASCFG(MI, NI, PF) {MI, NI, PF, "as" #MI "x" #NI "p" #PF, igemm_as_kernel<MI, NI, PF, false>, igemm_as_kernel<MI, NI, PF, true>}
const AsCfg kAsCfgs[] = {
    ASCFG(8, 2, 7), ASCFG(8, 2, 8), ASCFG(4, 2, 7), ASCFG(4, 2, 8), ASCFG(4, 4, 7), ASCFG(4, 4, 8),
    ASCFG(2, 2, 7), ASCFG(2, 2, 8), ASCFG(2, 4, 7), ASCFG(2, 4, 8), ASCFG(1, 4, 7), ASCFG(1, 4, 8),
};
constexpr int kNumAsCfgs = sizeof(kAsCfgs) / sizeof(kAsCfgs[0]);

}  // namespace

int astat_num_configs() { return kNumAsCfgs; }
const char* astat_config_name(int c) { return (c >= 0 && c < kNumAsCfgs) ? kAsCfgs[c].name : "?"; }
int astat_config_ni(int c) { return (c >= 0 && c < kNumAsCfgs) ? kAsCfgs[c].ni : 0; }

// Geometry of configuration `c` for the launch in `a` (filled by conv_impl): false = outside the family's domain.
// ppw_req: output-column passes per workgroup (0 = enough workgroups to fill the chip once).
bool astat_plan(const upk_ctx* ctx, const IgemmArgs& a, int c, int ppw_req, AsPlan* pl) {
  if (c < 0 || c >= kNumAsCfgs) return false;
  const AsCfg& k = kAsCfgs[c];
  if (!a.linear || a.ph_on || a.partial) return false;
  if (a.nchunks < k.pf || a.nchunks % k.pf) return false;
  if (a.npad % (k.ni * 16)) return false;  // (a wave's NI fragments are addressed off one base: all in range or none)
  const bool geglu = a.flags & UPK_F_GEGLU;
  // (the two-fragment GEGLU epilogue is the straight-line one: bias -> v * gelu(g) -> fp16)
  const bool geglu_plain = (a.flags & (UPK_F_GEGLU | UPK_F_SILU | UPK_F_QUICKGELU | UPK_F_OUT_F32 | UPK_F_OUT_NCHW_F32)) == UPK_F_GEGLU &&
                           !a.vt && !a.rowvec && !a.res && !(a.n_out & 3);
  if (geglu && k.ni == 2 && !geglu_plain) return false;
  const int BM = k.mi * 16, PW = AS_NW * k.ni * 16;
  const size_t lds = (size_t)a.nchunks * BM * 64 + (size_t)BM * 8 + (size_t)AS_NW * k.ni * 32 * 4;
  if (lds > 160 * 1024) return false;
  pl->bm = BM;
  pl->pw = PW;
  pl->lds_bytes = (int)lds;
  pl->npass = (a.npad + PW - 1) / PW;
  pl->tiles_m = (a.M + BM - 1) / BM;
  int ppw = ppw_req;
  if (ppw <= 0) {  // enough workgroups for one round of the chip, as few tile reloads as that allows
    const int want_n = (ctx->num_cus + pl->tiles_m - 1) / pl->tiles_m;
    const int tn = want_n < pl->npass ? (want_n < 1 ? 1 : want_n) : pl->npass;
    ppw = (pl->npass + tn - 1) / tn;
  }
  if (ppw > pl->npass) ppw = pl->npass;
  pl->ppw = ppw;
  pl->tiles_n = (pl->npass + ppw - 1) / ppw;
  pl->lds_bytes += 2 * ppw * PW * 4;  // bias | column sums of the workgroup's columns
  if (pl->lds_bytes > 160 * 1024) return false;
  return true;
}

int astat_launch(upk_ctx* ctx, IgemmArgs& a, int c, const AsPlan& pl, dim3 grid, hipStream_t stream) {
  const AsCfg& k = kAsCfgs[c];
  static unsigned long long attr_set[kNumAsCfgs][2] = {};
  if (int rc = upk_lds_attr_once(ctx, (const void*)k.fn, &attr_set[c][0])) return rc;
  if (int rc = upk_lds_attr_once(ctx, (const void*)k.fn_gen, &attr_set[c][1])) return rc;
  const bool geglu = a.flags & UPK_F_GEGLU;
  const bool fast = geglu ? Epi::plain_geglu(a) : (Epi::plain(a) && !a.gn_cp && !a.lnr_out && !a.rowvec);
  a.as_ppw = pl.ppw;
  a.as_npass = pl.npass;
  AsArgs s;
  memset(&s, 0, sizeof(s));
  s.x1 = a.x1, s.x2 = a.x2, s.x3 = a.x3, s.x4 = a.x4, s.w = a.w, s.zero = a.zero, s.bias = a.bias, s.res = a.res;
  s.y = (f16*)a.y, s.ln_u = a.ln_u, s.lnr_in = a.lnr_in;
  s.e1 = a.c1, s.e2 = a.c1 + a.c2, s.e3 = a.c1 + a.c2 + a.c3;
  s.ld1 = a.ld1, s.ld2 = a.ld2, s.ld3 = a.ld3, s.ld4 = a.ld4;
  s.M = a.M, s.npad = a.npad, s.n_out = a.n_out, s.nch = a.nchunks, s.flags = a.flags, s.ldr = a.ldr, s.ldy = a.ldy;
  s.ppw = pl.ppw, s.npass = pl.npass;
  s.tiles_m = a.tiles_m, s.tiles_n = a.tiles_n, s.xm_pm = a.xm_pm, s.xm_pn = a.xm_pn, s.xm_mi = a.xm_mi, s.xm_nj = a.xm_nj;
  s.lnr_slots = a.lnr_slots, s.ln_inv_dim = a.ln_inv_dim, s.ln_eps = a.ln_eps;
  hipLaunchKernelGGL(fast ? k.fn : k.fn_gen, grid, dim3(512), (size_t)pl.lds_bytes, stream, s, a);
  return upk_check_launch(ctx, "igemm_as");
}

}  // namespace upkd

// Implicit-GEMM convolution / Linear for gfx950 (CDNA4), fp16 in, fp32 accumulate.
//
//   y[m, n] = epilogue( sum_k A[m, k] * W[n, k] )
//   m = (b, oy, ox)  over the output pixels (token-major NHWC)
//   k = (tap, ci)    tap = ky*ks+kx, ci over the (possibly two-source) input channels
//
// Mapping to the hardware:
//   * v_mfma_f32_16x16x32_f16 (8 fp16 per lane per operand, 4 fp32 acc per lane),
//     operands SWAPPED (weights = A operand, activations = B operand) so that a lane
//     ends up with 4 consecutive output channels of one pixel -> 8-byte NHWC stores.
//   * K is walked in 32-wide chunks that never straddle a tap (channels are padded to
//     32), so the im2col address of a chunk is one scalar tap offset + a per-row pixel.
//   * global -> registers -> LDS double buffering, one barrier per K chunk; LDS tiles are
//     [rows][32 fp16] with a 16-byte-chunk XOR swizzle that makes ds_read_b128 of the
//     MFMA fragments conflict free (see lds_swz()).
//   * weights are pre-packed [K/32][n_pad][32] so a B tile is ONE contiguous run.
//   * split-K over blockIdx.z with partial slabs in caller workspace (fp32 accumulation inside a slice, the partial
//     rounded to fp16 — saturating — on its way out, summed in fp32 in slice order) and a
//     deterministic reduce+epilogue kernel (deep UNet levels have M = 96..384 rows only).
//
// Replaces F.conv2d / F.linear at the call sites listed in include/upk.h.
#include "igemm_common.h"

namespace {
using namespace upkd;


// KS = K-chunks (of 32) staged per barrier.  UNet launches have only 1-4 workgroups per CU,
// so latency must be hidden INSIDE a workgroup: KS chunks of global loads are in flight
// at once and KS*MI*NI MFMAs run between two barriers.
template <int MI, int NI, int WM, int WN, int KS>
__global__ __launch_bounds__(WM* WN * 64) void igemm_kernel(const IgemmArgs a) {
  constexpr int BM = MI * 16 * WM;
  constexpr int BN = NI * 16 * WN;
  constexpr int NT = WM * WN * 64;
  constexpr int A_IT = (BM * 4 + NT - 1) / NT;
  constexpr int B_IT = (BN * 4 + NT - 1) / NT;
  constexpr int A_TILE = BM * 32;  // halfs per K-chunk
  constexpr int B_TILE = BN * 32;
  constexpr int A_STAGE = KS * A_TILE;
  constexpr int B_STAGE = KS * B_TILE;

  __shared__ __attribute__((aligned(16))) f16 smem[2 * (A_STAGE + B_STAGE)];
  f16* sA = smem;                // [2][KS][A_TILE]
  f16* sB = smem + 2 * A_STAGE;  // [2][KS][B_TILE]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave - wm * WN;
  const int lg = lane >> 4;  // k-chunk group 0..3
  const int lc = lane & 15;

  // tile coordinates: consecutive blocks walk M first (they share the weight tile in L2)
  int tm, tn, zs;
  if (!tile_map(a, tm, tn, zs)) return;  // (whole workgroup, before any barrier)
  const int m0 = tm * BM;
  const int n0 = tn * BN;
  const int kc0 = zs * a.chunks_per_split;
  const int kc1 = min(a.nchunks, kc0 + a.chunks_per_split);

  const int ph = ph_id(a);  // (upsample phase launches: grid.y)
  // ---- per-thread A rows (im2col pixel decode, done once) ----
  int a_oy[A_IT], a_ox[A_IT], a_b[A_IT];
  bool a_ok[A_IT];
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int q = tid + i * NT;
    const int row = q >> 2;
    const int m = m0 + row;
    a_ok[i] = (row < BM) && (m < a.M);
    const int mm = a_ok[i] ? m : 0;
    if (a.linear) {  // 1x1 / Linear: the "pixel" is the row itself (no integer divisions)
      a_b[i] = 0;
      a_oy[i] = 0;
      a_ox[i] = mm;
    } else {
      const int b = div_hw(a, mm, HoWo);
      const int p = mm - b * HoWo;
      const int oy = div_w(a, p);
      a_b[i] = b;
      a_oy[i] = oy * a.stride - (a.ph_on ? 1 - (ph >> 1) : a.pad_lo);
      a_ox[i] = (p - oy * a.Wo) * a.stride - (a.ph_on ? 1 - (ph & 1) : a.pad_lo);
    }
  }

  f16x8 ra[KS][A_IT], rb[KS][B_IT];
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  // ---- K cursor: next chunk to load <-> (ky, kx, c0).  Kept incrementally (no integer
  // divisions in the loop) and the per-row pixel pointers are recomputed only when the
  // filter tap changes: the address math was costing ~14 VALU+SALU issues per MFMA.
  const int ctot = a.c1 + a.c2;
  int cur_kc = kc0, cur_c0, cur_ky, cur_kx;
  {
    const int tap = kc0 / a.cpt;
    cur_c0 = (kc0 - tap * a.cpt) * 32;
    cur_ky = tap / a.ks;
    cur_kx = tap - cur_ky * a.ks;
  }
  const f16* ap1[A_IT];
  const f16* ap2[A_IT];
  bool tap_ok[A_IT];
  auto set_tap = [&](int ky, int kx) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int ch = (tid + i * NT) & 3;
      int iy = a_oy[i] + ky;
      int ix = a_ox[i] + kx;
      tap_ok[i] = a_ok[i] && (a.linear || (iy >= 0 && iy < a.HL && ix >= 0 && ix < a.WL));
      if (a.ups) {
        iy >>= 1;
        ix >>= 1;
      }
      const long pix = tap_ok[i] ? ((long)a_b[i] * a.HS + iy) * a.WS + ix : 0;
      ap1[i] = a.x1 + pix * a.ld1 + ch * 8;
      ap2[i] = a.x2 ? a.x2 + pix * a.ld2 + ch * 8 - a.c1 : a.x1;
    }
  };
  set_tap(cur_ky, cur_kx);
  bool b_ok[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int row = (tid + i * NT) >> 2;
    b_ok[i] = (row < BN) && (n0 + row < a.npad);
  }
  const f16* wcur = a.w + (long)ph * a.ph_wstride + ((long)kc0 * a.npad + n0) * 32 + tid * 8;  // + i*NT*8 per B_IT
  const long wstep = (long)a.npad * 32;

  // issues the global loads of the next KS K-chunks (chunks >= kc1 read as zero)
  auto load_tiles = [&]() {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bool live = cur_kc < kc1;
      const bool second = cur_c0 >= a.c1;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const f16* p = (second ? ap2[i] : ap1[i]) + cur_c0;
        ra[s][i] = (live && tap_ok[i]) ? *(const f16x8*)p : zero8;
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i)
        rb[s][i] = (live && b_ok[i]) ? *(const f16x8*)(wcur + i * (NT * 8)) : zero8;
      wcur += wstep;
      ++cur_kc;
      cur_c0 += 32;
      if (cur_c0 == ctot) {
        cur_c0 = 0;
        if (++cur_kx == a.ks) {
          cur_kx = 0;
          ++cur_ky;
        }
        if (cur_kc < kc1) set_tap(cur_ky, cur_kx);
      }
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      f16* dA = sA + buf * A_STAGE + s * A_TILE;
      f16* dB = sB + buf * B_STAGE + s * B_TILE;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int q = tid + i * NT;
        const int row = q >> 2;
        if (row < BM) *(f16x8*)(dA + row * 32 + lds_swz(row, q & 3) * 8) = ra[s][i];
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int q = tid + i * NT;
        const int row = q >> 2;
        if (row < BN) *(f16x8*)(dB + row * 32 + lds_swz(row, q & 3) * 8) = rb[s][i];
      }
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment offset of this lane inside a 16-row sub-tile (swizzle depends on row&15 only)
  const int frag_off = lc * 32 + lds_swz(lc, lg) * 8;
  const int a_base = wm * (MI * 16) * 32 + frag_off;
  const int b_base = wn * (NI * 16) * 32 + frag_off;

  if (kc0 < kc1) {
    load_tiles();
    store_tiles(0);
  }
  __syncthreads();
  int cur = 0;
  for (int kc = kc0; kc < kc1; kc += KS) {
    const bool more = (kc + KS < kc1);
    if (more && !ABL_ON(ABL_NOGLOAD)) load_tiles();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const f16* tA = sA + cur * A_STAGE + s * A_TILE + a_base;
      const f16* tB = sB + cur * B_STAGE + s * B_TILE + b_base;
      f16x8 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = *(const f16x8*)(tA + i * 512);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = *(const f16x8*)(tB + j * 512);
      if (!ABL_ON(ABL_NOMFMA)) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(fb[j]));
      }
    }
    if (more && !ABL_ON(ABL_NOLDSW)) store_tiles(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: lane (lg, lc) holds rows m = .. + lc, cols n = .. + 4*lg + r ----
  const int mw = m0 + wm * (MI * 16);
  const int nw = n0 + wn * (NI * 16);
  if ABL_ON(ABL_NOEPI) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 123.456f) ((float*)a.y)[0] = t;
    return;
  }
  if (a.partial) {
    slab_t* slab = (slab_t*)a.partial + ((long)zs * (a.ph_on ? 4 : 1) + ph) * a.M * a.npad;  // uniform base
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = mw + i * 16 + lc;
      if (m >= a.M) continue;
      const unsigned roff = (unsigned)m * (unsigned)a.npad;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = nw + j * 16 + lg * 4;
        if (n < a.npad && !ABL_ON(ABL_NOSLAB)) slab_store(slab + roff + n, acc[i][j]);
      }
    }
    return;
  }
  if (a.lnr_in) Epi::lnr_fix<MI, NI>(a, mw, nw, lc, lg, acc);  // folded LayerNorm, row sums from the producer
  Epi::tile<MI, NI, WM, WN>(a, m0, mw, nw, lc, lg, acc, wm, wn, (float*)smem, a.M);
}


// ---------------------------------------------------------------------------------------
// Wave-specialised variant: waves 0..3 = MFMA consumers (same WM x WN tile decomposition as
// igemm_kernel), waves 4..7 = loaders that stream the A (im2col) and B (weight) tiles global
// -> LDS with direct-to-LDS DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write phase).
// LDS is a ring of NBUF stage slots; the loaders run D = NBUF-1 stages ahead and wait with a
// COUNTED vmcnt; one raw s_barrier per stage is the only synchronisation.  The consumers'
// instruction stream is ds_read + MFMA only, so load issue / latency / LDS fill overlap the
// matrix work inside ONE workgroup — what the UNet's 64..512-workgroup launches cannot get
// from co-resident workgroups.  LDS-DMA writes lane-linear (base + lane*16 B), so the XOR
// swizzle is applied to the SOURCE chunk a lane fetches; padding / out-of-range rows fetch
// from a zero page.
template <int P>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
}

// APP: the loader also walks the appended 1x1 K segment (IgemmArgs::x3 | x4).  A separate instantiation: the loader
// waves set the fill rate of the K loop, and even the few extra scalar registers of the switch-over cost 1.8 % of a
// UNet forward when they sat in the common kernels.
// LNF: the MFMA waves also take the folded LayerNorm's row statistics (IgemmArgs::ln_u) — likewise its own
// instantiation (M x N split configurations only): a wave-uniform test per K chunk and 2*MI live registers less in
// the kernels every other launch uses.
// LW: loader waves (4, or 8 in the "...l8" configurations of round 6: the loaders' DMA ISSUE — each global_load_lds holds its
// wave 60-185 cycles — is what bounds the K loop's fill at ~29 B/clk per CU; twice the issuers on the same ring.  Loader
// waves 4 .. LW-1 take no part in the shared epilogue and leave after the K loop.)
template <int MI, int NI, int WM, int WN, int KS, int NBUF, bool APP = false, bool LNF = false, int LW = 4>
__global__ __launch_bounds__(256 + LW * 64) void igemm_ws_kernel(const IgemmArgs a) {
  // WM*WN == 4: the 4 MFMA waves tile the block in M x N (each a MI x NI register tile).
  // WM*WN == 1: K-SPLIT mode — every MFMA wave owns the WHOLE MI x NI block tile and takes every
  // 4th K-chunk; the four accumulators are summed through LDS once at the end (fixed order).
  // Small block tiles (more workgroups for the 256 CUs) then keep a square-ish register tile:
  // LDS read traffic per MFMA is (MI+NI)/(MI*NI) KiB instead of the same figure for a
  // 4x smaller wave tile — the M x N split of a 64x112 block is LDS-bandwidth bound.
  static_assert(WM * WN == 4 || WM * WN == 1, "4 consumer waves");
  constexpr bool KSPLIT = (WM * WN == 1);
  static_assert(!KSPLIT || KS % 4 == 0, "K-split needs KS % 4 == 0");
  constexpr int BM = MI * 16 * WM;
  constexpr int BN = NI * 16 * WN;
  constexpr int AG = BM / 16, BG = BN / 16;          // 16-row groups (1 KiB = one wave DMA)
  static_assert(LW == 4 || (LW == 8 && !KSPLIT), "loader waves");
  constexpr int AGW = (AG + LW - 1) / LW, BGW = (BG + LW - 1) / LW;  // groups per loader wave
  constexpr int P = KS * (AGW + BGW);                 // DMAs per loader wave per stage
  constexpr int D = NBUF - 1;                         // prefetch distance in stages
  static_assert(D * P <= 63, "vmcnt range");
  constexpr int ROWS = BM + BN;
  constexpr int STAGE = KS * ROWS * 32;  // halfs per ring slot
  constexpr int DUMP = 512;              // 1 KiB dump row group for balance DMAs

  __shared__ __attribute__((aligned(16))) f16 smem[NBUF * STAGE + DUMP];
  // Shared epilogue of the M x N-split configurations (plain epilogue, no slabs): an MFMA wave keeps the first NJ0
  // column fragments of its register tile and hands the other NJ1 to the loader wave beside it through LDS — operand
  // round trip, stores and GroupNorm partials of a tile run on eight waves instead of four (in-kernel stamps of the
  // 3x3 224 -> 224 conv: epilogue 10.5-13.4 k of 43.7 k cycles on four waves; the K-split kernels' fragment exchange
  // already lets every wave finish any fragment)
  constexpr int NJ0 = (NI + 1) / 2, NJ1 = NI - NJ0;
  constexpr int HAND_OFF = 8192;  // bytes: behind tile_plain_cp's scratch (WM * WN * NI * 32 floats <= 4 KiB)
#ifdef UPK_KSPLIT_EPI4
  constexpr bool SHARE = false;
#else
  constexpr bool SHARE = !KSPLIT && NJ1 >= 1 && HAND_OFF + 4 * MI * NJ1 * 1024 <= (NBUF * STAGE + DUMP) * 2;
#endif
  static_assert(WM * WN * NI * 32 * 4 <= HAND_OFF, "tile_plain_cp scratch");

  if ABL_ON(ABL_EMPTY) return;
  const bool tl = ABL_ON(ABL_TIMELINE) && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && blockIdx.z == 0;
  unsigned long long* tlp = a.dbg + (blockIdx.x == 0 ? 0 : 32);
#ifdef UPK_TIMELINE
#define STAMP(i) do { if (tl && (threadIdx.x & 63) == 0) tlp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { (void)tl; (void)tlp; } while (0)
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tm, tn, zs;
  if (!tile_map(a, tm, tn, zs)) return;  // (whole workgroup, before any barrier)
  const int m0 = tm * BM;
  const int n0 = tn * BN;
  const int kc0 = zs * a.chunks_per_split;
  const int kc1 = min(a.nchunks, kc0 + a.chunks_per_split);
  const int nstages = (kc1 - kc0 + KS - 1) / KS;
  const int ph = ph_id(a);  // (upsample phase launches: grid.y)

  if (wave >= 4) {
    // ================================ loader ================================
    const int lw = wave - 4;
    if (lw == 0) STAMP(8);
    const int r16 = lane >> 2;                                  // row inside a 16-row group
    const int chd = (lane & 3) ^ ((-(lane >> 4)) & 3);          // source chunk of this lane (swizzle)
    const f16* zsrc = a.zero + (lane & 3) * 8;
    // A rows of this lane (one per owned group)
    int a_oy[AGW], a_ox[AGW], a_b[AGW];
    bool a_ok[AGW];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < AGW; ++i) {
      const int rg = lw + LW * i;
      const int m = m0 + rg * 16 + r16;
      a_ok[i] = (rg < AG) && (m < a.M);
      const int mm = a_ok[i] ? m : 0;
      if (a.linear) {  // 1x1 / Linear: the "pixel" is the row itself (no integer divisions)
        a_b[i] = 0;
        a_oy[i] = 0;
        a_ox[i] = mm;
      } else {
        const int b = div_hw(a, mm, HoWo);
        const int p = mm - b * HoWo;
        const int oy = div_w(a, p);
        a_b[i] = b;
        a_oy[i] = oy * a.stride - (a.ph_on ? 1 - (ph >> 1) : a.pad_lo);
        a_ox[i] = (p - oy * a.Wo) * a.stride - (a.ph_on ? 1 - (ph & 1) : a.pad_lo);
      }
    }
    int cur_kc = kc0, cur_c0, cur_ky, cur_kx;
    const f16* ap1[AGW];
    const f16* ap2[AGW];
    bool tap_ok[AGW];
    // APP: the current source pair is (x1 | x2) over the ks x ks taps, then (x3 | x4) at the output pixel
    bool in_app = false;
    auto set_tap = [&](int ky, int kx) {
      const f16* sx1 = (APP && in_app) ? a.x3 : a.x1;
      const f16* sx2 = (APP && in_app) ? a.x4 : a.x2;
      const int sc1 = (APP && in_app) ? a.c3 : a.c1;
      const int sld1 = (APP && in_app) ? a.ld3 : a.ld1;
      const int sld2 = (APP && in_app) ? a.ld4 : a.ld2;
#pragma unroll
      for (int i = 0; i < AGW; ++i) {
        int iy = a_oy[i] + ky;
        int ix = a_ox[i] + kx;
        tap_ok[i] = a_ok[i] && (a.linear || (iy >= 0 && iy < a.HL && ix >= 0 && ix < a.WL));
        if (a.ups) {
          iy >>= 1;
          ix >>= 1;
        }
        const long pix = tap_ok[i] ? ((long)a_b[i] * a.HS + iy) * a.WS + ix : 0;
        ap1[i] = sx1 + pix * sld1 + chd * 8;
        ap2[i] = sx2 ? sx2 + pix * sld2 + chd * 8 - sc1 : sx1;
      }
    };
    int sc1 = a.c1;            // channels of the first source of the current pair
    int ctot = a.c1 + a.c2;    // channels of the current pair
    auto enter_append = [&]() {  // stride 1, no upsample: tap (pad_lo, pad_lo) is the output pixel itself
      in_app = true;
      sc1 = a.c3;
      ctot = a.c3 + a.c4;
      set_tap(a.pad_lo, a.pad_lo);
    };
    if (APP && kc0 >= a.nchunks_main) {
      cur_c0 = (kc0 - a.nchunks_main) * 32;
      cur_ky = cur_kx = 0;
      enter_append();
    } else {
      const int tap = kc0 / a.cpt;
      cur_c0 = (kc0 - tap * a.cpt) * 32;
      cur_ky = tap / a.ks;
      cur_kx = tap - cur_ky * a.ks;
      set_tap(cur_ky, cur_kx);
    }
    // B rows of this lane
    const f16* bp[BGW];
    bool b_ok[BGW];
#pragma unroll
    for (int i = 0; i < BGW; ++i) {
      const int rg = lw + LW * i;
      const int row = rg * 16 + r16;
      b_ok[i] = (rg < BG) && (n0 + row < a.npad);
      bp[i] = a.w + (long)ph * a.ph_wstride + ((long)kc0 * a.npad + n0 + row) * 32 + chd * 8;
    }
    const long wstep = (long)a.npad * 32;

    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    // issues the P DMAs of the next stage into ring slot `slot`
    auto issue_stage = [&](int slot) {
      f16* base = smem + slot * STAGE;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bool live = cur_kc < kc1;
        const bool second = cur_c0 >= sc1;
#pragma unroll
        for (int i = 0; i < AGW; ++i) {
          const int rg = lw + LW * i;  // wave-uniform
          const f16* src = (live && tap_ok[i] && !ABL_ON(ABL_NOA)) ? ((second ? ap2[i] : ap1[i]) + cur_c0) : zsrc;
          f16* dst = (rg < AG) ? base + (s * ROWS + rg * 16) * 32 : smem + NBUF * STAGE;
          __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)dst, 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BGW; ++i) {
          const int rg = lw + LW * i;
          const f16* src = (live && b_ok[i] && !ABL_ON(ABL_NOB)) ? bp[i] : zsrc;
          f16* dst = (rg < BG) ? base + (s * ROWS + BM + rg * 16) * 32 : smem + NBUF * STAGE;
          // (nt on the weight stream of the split-K launches — each line read once by one or two workgroups — measured
          // +70 us on the forward, 2.946 -> 3.015 ms: the slices ARE shared across the XCD's M tiles through L2)
          __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)dst, 16, 0, 0);
          bp[i] += wstep;
        }
        ++cur_kc;
        cur_c0 += 32;
        if (cur_c0 == ctot) {
          cur_c0 = 0;
          if (++cur_kx == a.ks) {
            cur_kx = 0;
            ++cur_ky;
          }
          if (cur_kc < kc1) {
            if (APP && cur_kc == a.nchunks_main) enter_append();
            else set_tap(cur_ky, cur_kx);
          }
        }
      }
    };
    // waits until at most `r - 1` whole stages are still in flight (r = stages outstanding)
    auto wait_oldest = [&](int r) {
      static_assert(D <= 7, "wait ladder");
      if (D >= 7 && r >= 7) wait_vmcnt<6 * P>();
      else if (D >= 6 && r == 6) wait_vmcnt<(D >= 6 ? 5 : 0) * P>();
      else if (D >= 5 && r == 5) wait_vmcnt<(D >= 5 ? 4 : 0) * P>();
      else if (D >= 4 && r == 4) wait_vmcnt<(D >= 4 ? 3 : 0) * P>();
      else if (D >= 3 && r == 3) wait_vmcnt<(D >= 3 ? 2 : 0) * P>();
      else if (r == 2) wait_vmcnt<P>();
      else wait_vmcnt<0>();
    };
    int issued = 0;
    if (lw == 0) STAMP(9);
    for (; issued < D && issued < nstages; ++issued) issue_stage(issued % NBUF);
    if (lw == 0) STAMP(10);
    wait_oldest(issued);  // stage 0 landed
    if (lw == 0) STAMP(11);
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < nstages; ++t) {
      // slot (t + D) % NBUF held stage t - 1: its readers passed the barrier that ended iteration t-1
      if (issued < nstages) {
        if (!ABL_ON(ABL_NOGLOAD)) issue_stage(issued % NBUF);
        ++issued;
      }
      const int outstanding = issued - (t + 1);  // stages t+1 .. issued-1
      if (outstanding > 0) wait_oldest(outstanding);
      __builtin_amdgcn_s_barrier();
    }
    if (lw == 0) STAMP(12);
    // K-split configurations: the loader waves stay for the epilogue — the four K slices meet in LDS anyway, and eight
    // waves finishing NF / 8 fragments each (operand round trip, stores, GroupNorm partials) are through in about half
    // the time four take (in-kernel stamps: epilogue 10-13 k of a 44 k-cycle launch)
#ifdef UPK_KSPLIT_EPI4  // (A/B builds: the four MFMA waves alone)
    return;
#else
    if constexpr (LW > 4) {
      if (lw >= 4) return;  // (a finished wave no longer counts at the workgroup's barriers)
    }
    if constexpr (!KSPLIT) {
      if constexpr (SHARE) {
        if ABL_ON(ABL_NOEPI) return;
        if (Epi::plain(a) && !a.partial && !a.lnr_out) {  // (workgroup-uniform: the MFMA waves take the same branch)
          const int pwm = lw / WN, pwn = lw - pwm * WN;   // the MFMA wave beside this one
          const int lg = lane >> 4, lc = lane & 15;
          __builtin_amdgcn_s_barrier();                   // its NJ1 fragments are in LDS
          const float* hand = (const float*)((const char*)smem + HAND_OFF) + lw * (MI * NJ1) * 256;
          f32x4 acc2[MI][NJ1];
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ1; ++j) acc2[i][j] = *(const f32x4*)(hand + ((i * NJ1 + j) * 64 + lane) * 4);
          const int mw = m0 + pwm * (MI * 16), nw = n0 + pwn * (NI * 16) + NJ0 * 16;
          if (a.gn_cp && !ABL_ON(ABL_NOGNP)) Epi::tile_plain_cp<MI, NJ1, WM, WN, NI>(a, m0, mw, nw, lc, lg, acc2, pwm, pwn, (float*)smem, a.M, NJ0);
          else Epi::tile_plain<MI, NJ1>(a, mw, nw, lc, lg, acc2, a.M);
        }
      }
      return;
    }
#endif
  }

  // ================================ consumers ================================
  if (wave == 0) STAMP(0);
#ifdef UPK_R6_EXPERIMENTS
  if (a.pf_self) {
    // (experiment, DESIGN.md 14h: dev builds only — it loses with four forwards in flight and on the latency table) this launch's OWN weight slice -> this XCD's L2, cooperatively: the workgroups of an XCD that
    // share the (N tile, K split) slice each touch every cnt-th chunk of it up front, so that the slice is requested from HBM
    // through cnt CUs' load windows at once instead of through each CU's own ~64 KB window, chunk after chunk
    int rank, cnt;
    if (a.xm_pm == 0) {  // default tile order: workgroup w = tile index runs on XCD w % 8, M tiles fastest
      cnt = max(1, a.tiles_m >> 3);
      rank = (tm >> 3) % cnt;
    } else {
      cnt = max(1, a.xm_mi);
      rank = (int)(blockIdx.x >> 3) % cnt;
    }
    const int lpc = BN / 2;  // 128-byte lines of one chunk's BN x 64 B block
    const int mine = (kc1 - kc0 - rank + cnt - 1) / cnt;  // chunks kc0 + rank, + cnt, ...
    const int total = mine * lpc;
    const int lines_ok = min(BN, a.npad - n0) / 2;  // (rows past n_pad do not exist)
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const f16* wb = a.w + (long)ph * a.ph_wstride + ((long)kc0 * a.npad + n0) * 32;
    for (int l0 = wave * 64; l0 < total; l0 += 256) {
      const int l = min(l0 + lane, total - 1);
      const int ci = l / lpc, li = min(l - ci * lpc, lines_ok - 1);
      const f16* src = wb + ((long)(rank + ci * cnt) * a.npad) * 32 + li * 64;
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(smem + NBUF * STAGE), 16, 0, 0);
    }
  }
#endif
  if (a.pf_lines > 0) {
    // next launch's weights -> memory-side cache (include/upk.h pf_next): this workgroup's share of the lines, one 16-byte
    // piece per line and lane, by direct-to-LDS loads into the dump row group.  The MFMA waves have nothing to do until ring
    // stage 0 lands; the loads return whenever HBM answers and nobody waits for their data.
    const int nwg = gridDim.x * gridDim.y * gridDim.z;
    const int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int per = (a.pf_lines + nwg - 1) / nwg;
    const int first = wg * per, last = min(a.pf_lines, first + per) - 1;
    if (first <= last) {
      typedef __attribute__((address_space(3))) void* lds_ptr;
      typedef const __attribute__((address_space(1))) void* glb_ptr;
      for (int l0 = first + wave * 64; l0 <= last; l0 += 256) {  // (wave-uniform trip count; lanes past the end re-touch the last line)
        const int l = min(l0 + lane, last);
        __builtin_amdgcn_global_load_lds((glb_ptr)(a.pf + (long)l * 128), (lds_ptr)(smem + NBUF * STAGE), 16, 0, 0);
      }
    }
  }
  const int wm = KSPLIT ? 0 : wave / WN;
  const int wn = KSPLIT ? 0 : wave - wm * WN;
  const int lg = lane >> 4;
  const int lc = lane & 15;
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int frag_off = lc * 32 + lds_swz(lc, lg) * 8;
  const int a_base = wm * (MI * 16) * 32 + frag_off;
  const int b_base = (BM + wn * (NI * 16)) * 32 + frag_off;

  // folded LayerNorm (M x N split only): per-row sum / sum of squares of the A fragments this wave
  // reads anyway; lane (lc, lg) sees row lc, k-slice lg of every chunk
  constexpr bool ln = LNF;
  float ln_s1[MI], ln_s2[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) ln_s1[i] = ln_s2[i] = 0.f;

  if (wave < 4) __builtin_amdgcn_s_barrier();  // stage 0 is in LDS
  if (wave == 0) STAMP(1);
  for (int t = 0; t < (wave < 4 ? nstages : 0); ++t) {
    if (wave == 0 && t == 1) STAMP(2);
    const f16* slot = smem + (t % NBUF) * STAGE;
#pragma unroll
    for (int s0 = 0; s0 < (KSPLIT ? KS / 4 : KS); ++s0) {
      const int s = KSPLIT ? wave + 4 * s0 : s0;  // K-split: this wave's chunks of the stage
      if ABL_ON(ABL_NOLDSW) continue;  // (ablation: no LDS reads either)
      const f16* tA = slot + s * ROWS * 32 + a_base;
      const f16* tB = slot + s * ROWS * 32 + b_base;
      f16x8 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = *(const f16x8*)(tA + i * 512);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = *(const f16x8*)(tB + j * 512);
      if (!ABL_ON(ABL_NOMFMA)) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        if constexpr (!KSPLIT) {
          if (ln) {  // VALU work beside the MFMAs above
            const f16x2 one2 = {(f16)1.f, (f16)1.f};
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
              for (int h = 0; h < 4; ++h) {
                const f16x2 xx = {fa[i][2 * h], fa[i][2 * h + 1]};
                ln_s1[i] = __builtin_amdgcn_fdot2(xx, one2, ln_s1[i], false);
                ln_s2[i] = __builtin_amdgcn_fdot2(xx, xx, ln_s2[i], false);
              }
            }
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(fb[j]));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // our LDS reads are done before the slot is reused
    __builtin_amdgcn_s_barrier();
  }
  if (wave == 0) STAMP(3);

  if constexpr (KSPLIT) {
    // ---- sum the 4 K-slices through LDS (the ring is free: all consumers passed the last barrier)
    constexpr int NF = MI * NI;
    static_assert(NF * 4096 <= NBUF * STAGE * 2, "reduction buffer must fit in the ring");
    float* red = (float*)smem;  // [4 waves][NF fragments][64 lanes][4]
    if (wave < 4) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) *(f32x4*)(red + ((wave * NF + i * NI + j) * 64 + lane) * 4) = acc[i][j];
    }
    __syncthreads();  // (all eight waves: the loaders come here from their last stage)
#ifdef UPK_KSPLIT_EPI4
    constexpr int EW = 4;
#else
    constexpr int EW = 8;  // waves in the epilogue
#endif
    if ABL_ON(ABL_NOEPI) return;
    auto frag_sum = [&](int f) {
      f32x4 v = *(const f32x4*)(red + ((0 * NF + f) * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) v += *(const f32x4*)(red + ((w * NF + f) * 64 + lane) * 4);
      return v;
    };
    slab_t* slab = a.partial ? (slab_t*)a.partial + ((long)zs * (a.ph_on ? 4 : 1) + ph) * a.M * a.npad : nullptr;
    if (slab) {
      for (int f = wave; f < NF; f += EW) {  // fragment f = (i, j) is finished by wave f % 8
        const int i = f / NI, j = f - i * NI;
        const int m = m0 + i * 16 + lc;
        const int n = n0 + j * 16 + lg * 4;
        if (n < a.npad && m < a.M && !ABL_ON(ABL_NOSLAB)) slab_store(slab + (unsigned)m * (unsigned)a.npad + n, frag_sum(f));
      }
      return;
    }
    // this wave's fragments f = wave + 8q
    constexpr int FW = (NF + EW - 1) / EW;
    if (Epi::plain(a)) {  // straight-line common case (see Epi::Plain)
      const Epi::Plain P(a);
      Epi::Plain::Row rows[FW];
      f32x4 bvs[FW], rvs[FW];
      f16x4 rrs[FW];
#pragma unroll
      for (int q = 0; q < FW; ++q) {
        const int f = wave + EW * q;
        const int i = f / NI, j = f - i * NI;
        const int n = n0 + j * 16 + lg * 4;
        rows[q] = P.row(a, f < NF ? m0 + i * 16 + lc : a.M, a.M);
        bvs[q] = P.bias4(a, n);
        rvs[q] = P.rv4(a, rows[q], n);
        rrs[q] = P.res4(a, rows[q], n);
      }
      // GroupNorm channel partials (a.gn_cp, see Epi::tile_plain_cp): per finished fragment the 16-row sums go to
      // LDS behind the K-slice buffer, then the MI row fragments of each column fragment are combined
      static_assert(NF * 4096 + NF * 128 <= NBUF * STAGE * 2, "partials must fit behind the reduction buffer");
      float* cpred = red + NF * 1024;  // [NF][2][16]
#pragma unroll
      for (int q = 0; q < FW; ++q) {
        const int f = wave + EW * q;
        if (f >= NF) continue;
        const int j = f - (f / NI) * NI;
        f32x4 fs = frag_sum(f);
        if (a.lnr_in) fs = Epi::lnr_fix1(a, m0 + (f / NI) * 16 + lc, n0 + j * 16 + lg * 4, fs);
        const f16x4 o = Epi::Plain::put(a, rows[q], n0 + j * 16 + lg * 4, fs + bvs[q] + rvs[q], rrs[q]);
        if (a.gn_cp) {
          f32x4 su, sq;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float v = (float)o[k];
            su[k] = Epi::row_sum16(v);
            sq[k] = Epi::row_sum16(v * v);
          }
          if (lc == 0) {
            *(f32x4*)(cpred + f * 32 + lg * 4) = su;
            *(f32x4*)(cpred + f * 32 + 16 + lg * 4) = sq;
          }
        } else if (a.lnr_out) {  // LayerNorm row sums of the output (IgemmArgs::lnr_out): this fragment's 16 columns
          float s1 = 0.f, s2 = 0.f;
          if (n0 + j * 16 + lg * 4 < a.n_out) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float v = (float)o[k];
              s1 += v;
              s2 += v * v;
            }
          }
          s1 = sum_lane_groups(s1);
          s2 = sum_lane_groups(s2);
          if (lg == 0) {
            cpred[f * 32 + lc] = s1;
            cpred[f * 32 + 16 + lc] = s2;
          }
        }
      }
      if (a.lnr_out) {  // (workgroup-uniform) one slot per N tile: the NI column fragments of each row combined
        __syncthreads();
        const int t = wave * 64 + lane;
        if (t < MI * 16) {
          const int i = t >> 4, r = t & 15;
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            s1 += cpred[(i * NI + j) * 32 + r];
            s2 += cpred[(i * NI + j) * 32 + 16 + r];
          }
          const int m = m0 + i * 16 + r;
          if (m < a.M) *(f32x2*)(a.lnr_out + ((long)(n0 / (NI * 16)) * a.M + m) * 2) = (f32x2){s1, s2};
        }
      }
      if (a.gn_cp) {  // (workgroup-uniform)
        __syncthreads();
        const int b = m0 / a.gn_hw;
        const int blk = (m0 - b * a.gn_hw) / (MI * 16);
        float* dst = a.gn_cp + (long)((b * a.gn_nblk + blk) * 2) * a.npad;
        if (lane < 32) {
          const int which = lane >> 4, col = lane & 15;
          for (int j = wave; j < NI; j += EW) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i) t += cpred[(i * NI + j) * 32 + which * 16 + col];
            const int n = n0 + j * 16 + col;
            if (n < a.npad) dst[which * a.npad + n] = t;
          }
        }
      }
      return;
    }
    const bool geglu = a.flags & UPK_F_GEGLU;
    RowCtx rcs[FW];
    Epi::In ins[FW];
    f32x4 bvs[FW], bgs[FW];
#pragma unroll
    for (int q = 0; q < FW; ++q) {
      const int f = wave + EW * q;
      const int i = f / NI, j = f - i * NI;
      const int n = n0 + j * 16 + lg * 4;
      const bool live = f < NF && !(geglu && (j & 2));
      rcs[q] = Epi::row(a, live ? m0 + i * 16 + lc : a.M, a.M);
      bvs[q] = Epi::bias4(a, live ? n : a.npad);
      bgs[q] = Epi::bias4(a, live && geglu ? n + 32 : a.npad);
      ins[q] = Epi::fetch(a, rcs[q], n);
    }
#pragma unroll
    for (int q = 0; q < FW; ++q) {
      const int f = wave + EW * q;
      if (f >= NF) continue;
      const int i = f / NI, j = f - i * NI;
      const int n = n0 + j * 16 + lg * 4;
      if (n >= a.npad) continue;
      if (geglu) {
        if ((j & 2) == 0 && j + 2 < NI) {
          f32x4 v = frag_sum(f), gte = frag_sum(f + 2);
          if (a.lnr_in) {
            v = Epi::lnr_fix1(a, m0 + i * 16 + lc, n, v);
            gte = Epi::lnr_fix1(a, m0 + i * 16 + lc, n + 32, gte);
          }
          Epi::store(a, rcs[q], n, v, gte, bvs[q], bgs[q], ins[q]);
        }
      } else {
        f32x4 v = frag_sum(f);
        if (a.lnr_in) v = Epi::lnr_fix1(a, m0 + i * 16 + lc, n, v);
        Epi::store(a, rcs[q], n, v, v, bvs[q], bvs[q], ins[q]);
      }
    }
    return;
  }
  const int mw = m0 + wm * (MI * 16);
  const int nw = n0 + wn * (NI * 16);
  if (ln) {  // y = rstd * (acc - mean * colsum): the lane's accumulator row is the row it has the statistics of
    f32x4 u[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = nw + j * 16 + lg * 4;
      u[j] = n < a.npad ? *(const f32x4*)(a.ln_u + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float s1 = ln_s1[i], s2 = ln_s2[i];
      s1 += __shfl_xor(s1, 16);
      s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      const float mean = s1 * a.ln_inv_dim;
      const float var = fmaxf(s2 * a.ln_inv_dim - mean * mean, 0.f);
      const float rstd = rsqrtf(var + a.ln_eps);
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = (acc[i][j] - mean * u[j]) * rstd;
    }
  } else if (a.lnr_in) {
    Epi::lnr_fix<MI, NI>(a, mw, nw, lc, lg, acc);
  }
  if ABL_ON(ABL_NOEPI) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 123.456f) ((float*)a.y)[0] = t;
    return;
  }
  if (a.partial) {
    slab_t* slab = (slab_t*)a.partial + ((long)zs * (a.ph_on ? 4 : 1) + ph) * a.M * a.npad;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = mw + i * 16 + lc;
      if (m >= a.M) continue;
      const unsigned roff = (unsigned)m * (unsigned)a.npad;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = nw + j * 16 + lg * 4;
        if (n < a.npad && !ABL_ON(ABL_NOSLAB)) slab_store(slab + roff + n, acc[i][j]);
      }
    }
    return;
  }
  if constexpr (SHARE) {
    if (Epi::plain(a) && !a.lnr_out) {  // (the loader beside this wave finishes fragments NJ0 .. NI - 1)
      float* hand = (float*)((char*)smem + HAND_OFF) + wave * (MI * NJ1) * 256;
      f32x4 acc1[MI][NJ0];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          if (j < NJ0) acc1[i][j] = acc[i][j];
          else *(f32x4*)(hand + ((i * NJ1 + (j - NJ0)) * 64 + lane) * 4) = acc[i][j];
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (a.gn_cp && !ABL_ON(ABL_NOGNP)) Epi::tile_plain_cp<MI, NJ0, WM, WN, NI>(a, m0, mw, nw, lc, lg, acc1, wm, wn, (float*)smem, a.M, 0);
      else Epi::tile_plain<MI, NJ0>(a, mw, nw, lc, lg, acc1, a.M);
#ifdef UPK_TIMELINE
      if (wave == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(4);
      }
#endif
      return;
    }
  }
  Epi::tile<MI, NI, WM, WN>(a, m0, mw, nw, lc, lg, acc, wm, wn, (float*)smem, a.M);
#ifdef UPK_TIMELINE
  if (wave == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(4);
  }
#endif
#undef STAMP
}

// Split-K second pass: sums the slabs in fixed order (deterministic) and runs the epilogue.
__global__ __launch_bounds__(256) void igemm_reduce_kernel(const IgemmArgs a, int splitk) {
  const int nq = a.npad >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)a.M * nq;
  if (idx >= total) return;
  const int m = (int)(idx / nq);
  const int n = (int)(idx - (long)m * nq) * 4;
  const long slab = (long)a.M * a.npad * (a.ph_on ? 4 : 1);  // (phase launches: slabs [z][phase], grid.y = phase)
  const slab_t* p = (const slab_t*)a.partial + ((long)ph_id(a) * a.M + m) * a.npad + n;
  f32x4 v = {0, 0, 0, 0}, g = {0, 0, 0, 0};
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  if (a.flags & UPK_F_GEGLU) {
    if (n & 32) return;  // gate columns are consumed by their value partner
    for (int z = 0; z < splitk; ++z) {
      v += slab_load4(p + z * slab);
      g += slab_load4(p + z * slab + 32);
    }
  } else {
    // four slabs in flight (a one-load-per-iteration loop with a runtime trip count runs at the latency of a load)
    for (int z0 = 0; z0 < splitk; z0 += 4) {
      f32x4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = z0 + u < splitk ? slab_load4(p + (long)(z0 + u) * slab) : z4;
#pragma unroll
      for (int u = 0; u < 4; ++u) v += t[u];  // (slab order, as before)
    }
  }
  IgemmArgs b = a;
  b.partial = nullptr;
  const RowCtx rc = Epi::row(b, m, b.M);
  Epi::store(b, rc, n, v, g, Epi::bias4(b, n), Epi::bias4(b, (a.flags & UPK_F_GEGLU) ? n + 32 : a.npad),
             Epi::fetch(b, rc, n));
}

// Split-K second pass that applies the GroupNorm (+ SiLU) of what it reduces (include/upk.h gno_*): one workgroup per
// (sample, group).  A thread owns up to NV vectors of V consecutive channels of the group; every slab load of a
// vector is issued before the first use, the reduced + fp16-rounded values stay in registers across the group
// reduction (fp64, fixed order), and the normalised tensor is written from them: the partials are read once and the
// GroupNorm launch behind a split-K conv disappears.
struct GnApply {
  const float* gamma;
  const float* beta;
  f16* yn;
  float eps;
  int ldn, silu, skip_y, cpg, hw;
};

template <int V>
struct VecIO {
  static __device__ __forceinline__ void ldf(const float* p, bool on, float (&o)[V]) {
    if constexpr (V == 4) {
      const f32x4 w = on ? *(const f32x4*)p : (f32x4){0.f, 0.f, 0.f, 0.f};
      o[0] = w[0], o[1] = w[1], o[2] = w[2], o[3] = w[3];
    } else if constexpr (V == 2) {
      const f32x2 w = on ? *(const f32x2*)p : (f32x2){0.f, 0.f};
      o[0] = w[0], o[1] = w[1];
    } else {
      o[0] = on ? *p : 0.f;
    }
  }
  static __device__ __forceinline__ void lds(const slab_t* p, bool on, float (&o)[V]) {  // (a split-K slab element)
#ifdef UPK_SLAB_F32
    ldf(p, on, o);
#else
    ldh(p, on, o);
#endif
  }
  static __device__ __forceinline__ void ldh(const f16* p, bool on, float (&o)[V]) {
    if constexpr (V == 4) {
      const f16x4 w = on ? *(const f16x4*)p : (f16x4){0, 0, 0, 0};
      o[0] = (float)w[0], o[1] = (float)w[1], o[2] = (float)w[2], o[3] = (float)w[3];
    } else if constexpr (V == 2) {
      const f16x2 w = on ? *(const f16x2*)p : (f16x2){0, 0};
      o[0] = (float)w[0], o[1] = (float)w[1];
    } else {
      o[0] = on ? (float)*p : 0.f;
    }
  }
  static __device__ __forceinline__ void sth(f16* p, const f16 (&v)[V]) {
    if constexpr (V == 4) {
      *(f16x4*)p = (f16x4){v[0], v[1], v[2], v[3]};
    } else if constexpr (V == 2) {
      *(f16x2*)p = (f16x2){v[0], v[1]};
    } else {
      *p = v[0];
    }
  }
};

template <int V, int NV>
__global__ __launch_bounds__(256) void igemm_reduce_gnapply_kernel(const IgemmArgs a, int splitk, const GnApply g) {
  using IO = VecIO<V>;
  __shared__ double red[8];
  int grp, b;
  upk_xcd_xb(grp, b);
  const int tid = threadIdx.x;
  const int vpr = g.cpg / V;
  const int nvec = g.hw * vpr;
  const int c0 = grp * g.cpg;
  const long slab = (long)a.M * a.npad;
  const int st = (a.rowvec && a.step) ? *a.step : 0;
  const float* rvb = a.rowvec ? a.rowvec + (unsigned)(st * a.rv_ss + b * a.rv_bs) : nullptr;
  float h[NV][V], ga[NV][V], be[NV][V];
  int mrow[NV], ncol[NV];
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = tid + k * 256;
    const bool on = i < nvec;
    const int p = on ? i / vpr : 0;
    const int n = c0 + (on ? i - p * vpr : 0) * V;
    const int m = b * g.hw + p;
    mrow[k] = on ? m : -1;
    ncol[k] = n;
    const slab_t* pp = (const slab_t*)a.partial + (long)m * a.npad + n;
    float acc[V], cb[V], cr[V], cs[V];
    IO::ldf(g.gamma + n, on, ga[k]);  // (not needed before the group reduction: in flight with the slabs)
    IO::ldf(g.beta + n, on, be[k]);
    IO::ldf(a.bias + n, on && a.bias != nullptr, cb);
    IO::ldf(rvb + n, on && rvb != nullptr, cr);
    IO::ldh(a.res + (long)m * a.ldr + n, on && a.res != nullptr, cs);
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int z0 = 0; z0 < splitk; z0 += 4) {
      float t[4][V];
#pragma unroll
      for (int zz = 0; zz < 4; ++zz) {
        const bool zon = on && z0 + zz < splitk;
        IO::lds(pp + (long)(zon ? z0 + zz : 0) * slab, zon, t[zz]);
      }
#pragma unroll
      for (int zz = 0; zz < 4; ++zz)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += t[zz][j];
    }
    f16 o[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      o[j] = (f16)(acc[j] + (cb[j] + cr[j]) + cs[j]);
      const float f = (float)o[j];
      h[k][j] = f;
      s += f;
      ss += f * f;
    }
    if (on && !g.skip_y) IO::sth((f16*)a.y + (long)m * a.ldy + n, o);
  }
  double ds = (double)s, dss = (double)ss;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    ds += __shfl_xor(ds, o);
    dss += __shfl_xor(dss, o);
  }
  if ((tid & 63) == 0) {
    red[(tid >> 6) * 2] = ds;
    red[(tid >> 6) * 2 + 1] = dss;
  }
  __syncthreads();
  const double n_el = (double)g.hw * g.cpg;
  const double mean = (((red[0] + red[2]) + red[4]) + red[6]) / n_el;
  double var = (((red[1] + red[3]) + red[5]) + red[7]) / n_el - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)g.eps));
  const float fmean = (float)mean;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const bool on = mrow[k] >= 0;
    const int n = ncol[k];
    f16 o[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float sc = rstd * ga[k][j];
      const float sh = be[k][j] - fmean * sc;
      float f = h[k][j] * sc + sh;
      if (g.silu) f = upk_silu(f);
      o[j] = (f16)f;
    }
    if (on) IO::sth(g.yn + (long)mrow[k] * g.ldn + n, o);
  }
}

// Split-K second pass that also takes the GroupNorm statistics of what it writes: grid (chunks, B) and
// thread layout of gn_stats_kernel (norm.hip) — a thread owns 8 consecutive channels, `rpi` pixels in
// flight — so the partial sums come out in upk_groupnorm's workspace layout and the GroupNorm that
// follows a split-K conv (ResBlock out_layers / the next block's in_layers) needs its apply pass only.
// Plain epilogue only (bias + timestep row vector + residual -> fp16 NHWC); statistics are taken from
// the fp16-rounded values, exactly what gn_stats_kernel would read back.
struct GnFuse {
  float* ws;
  int groups, cpg, hw, nchunks, pix_per_chunk;
};

__global__ __launch_bounds__(256) void igemm_reduce_gn_kernel(const IgemmArgs a, int splitk, const GnFuse gf) {
  __shared__ float sc[2][256 * 8];
  const int C = a.n_out;
  const int vpr = C >> 3;
  int chunk, b;
  upk_xcd_xb(chunk, b);
  const int tid = threadIdx.x;
  const int p0 = chunk * gf.pix_per_chunk;
  const int p1 = min(gf.hw, p0 + gf.pix_per_chunk);
  const int rpi = 256 / vpr;
  const int r = tid / vpr;
  const int v = tid - r * vpr;
  if (r < rpi) {
    const int n = v * 8;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 c0 = z4, c1 = z4;  // per-column constants: bias + this sample's timestep row vector
    if (a.bias) {
      c0 = *(const f32x4*)(a.bias + n);
      c1 = *(const f32x4*)(a.bias + n + 4);
    }
    if (a.rowvec) {
      const int st = a.step ? *a.step : 0;
      const float* rv = a.rowvec + (unsigned)(st * a.rv_ss + b * a.rv_bs) + n;
      c0 += *(const f32x4*)rv;
      c1 += *(const f32x4*)(rv + 4);
    }
    float s[8], ss[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
    const long slab = (long)a.M * a.npad;
    for (int p = p0 + r; p < p1; p += rpi) {
      const long m = (long)b * gf.hw + p;
      const slab_t* pp = (const slab_t*)a.partial + m * a.npad + n;
      f32x4 v0 = c0, v1 = c1;
#pragma unroll 4
      for (int z = 0; z < splitk; ++z) {
        v0 += slab_load4(pp + z * slab);
        v1 += slab_load4(pp + z * slab + 4);
      }
      if (a.res) {
        const f16x8 rr = *(const f16x8*)(a.res + m * a.ldr + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v0[j] += (float)rr[j];
          v1[j] += (float)rr[4 + j];
        }
      }
      f16x8 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = (f16)v0[j];
        o[4 + j] = (f16)v1[j];
      }
      *(f16x8*)((f16*)a.y + m * a.ldy + n) = o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)o[j];
        s[j] += f;
        ss[j] += f * f;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[0][r * C + v * 8 + j] = s[j];
      sc[1][r * C + v * 8 + j] = ss[j];
    }
  }
  __syncthreads();
  // fold (row slot, channel) -> groups in the fixed order of gn_stats_kernel
  const int q = tid >> 2, sub = tid & 3;
  float t = 0.f;
  if (q < gf.groups * 2) {
    const int g = q >> 1, which = q & 1;
    const int nel = rpi * gf.cpg;
    for (int e = sub; e < nel; e += 4) {
      const int rr = e / gf.cpg;
      t += sc[which][rr * C + g * gf.cpg + (e - rr * gf.cpg)];
    }
  }
  const float t1 = t + __shfl_xor(t, 1);
  const float t2 = t1 + __shfl_xor(t1, 2);
  if (q < gf.groups * 2 && sub == 0) gf.ws[((long)(b * gf.nchunks + chunk) * gf.groups) * 2 + q] = t2;
}

struct CfgInfo {
  int mi, ni, wm, wn, ks;
  const char* name;
  void (*fn)(const IgemmArgs);
  int nbuf;  // 0: classic register-staged kernel; > 0: wave-specialised DMA kernel (512 threads)
  void (*fn_app)(const IgemmArgs);  // variant whose loader walks an appended 1x1 K segment, or nullptr
  void (*fn_ln)(const IgemmArgs);   // variant whose MFMA waves take the folded LayerNorm's row statistics, or nullptr
  int lw = 4;                       // loader waves of the wave-specialised kernel (threads = 256 + 64 lw)
};

template <int MI, int NI, int WM, int WN, int KS, int NB>
constexpr void (*ws_ln_fn())(const IgemmArgs) {
  if constexpr (WM * WN > 1) return igemm_ws_kernel<MI, NI, WM, WN, KS, NB, false, true>;
  else return nullptr;
}

#define CFG(MI, NI, WM, WN, KS) \
  {MI, NI, WM, WN, KS, #MI "x" #NI "x" #WM "x" #WN "k" #KS, igemm_kernel<MI, NI, WM, WN, KS>, 0, nullptr, nullptr}
#define CFGW(MI, NI, WM, WN, KS, NB) \
  {MI, NI, WM, WN, KS, #MI "x" #NI "x" #WM "x" #WN "k" #KS "w" #NB, igemm_ws_kernel<MI, NI, WM, WN, KS, NB>, NB, \
   igemm_ws_kernel<MI, NI, WM, WN, KS, NB, true>, ws_ln_fn<MI, NI, WM, WN, KS, NB>(), 4}
// eight loader waves (768 threads); plain and appended-segment loaders, no fragment-side LayerNorm fold
#define CFGW8(MI, NI, WM, WN, KS, NB) \
  {MI, NI, WM, WN, KS, #MI "x" #NI "x" #WM "x" #WN "k" #KS "w" #NB "l8", igemm_ws_kernel<MI, NI, WM, WN, KS, NB, false, false, 8>, NB, \
   igemm_ws_kernel<MI, NI, WM, WN, KS, NB, true, false, 8>, nullptr, 8}
// (MI, NI, WM, WN, KS): block tile = (MI*16*WM) x (NI*16*WN), WM*WN waves, KS K-chunks/stage.
const CfgInfo kCfgs[] = {
    CFG(4, 4, 2, 2, 1), CFG(4, 4, 2, 2, 2),  // 128x128
    CFG(2, 4, 2, 2, 1), CFG(2, 4, 2, 2, 2),  //  64x128
    CFG(4, 2, 2, 2, 1), CFG(4, 2, 2, 2, 2),  // 128x64
    CFG(2, 2, 2, 2, 1), CFG(2, 2, 2, 2, 4),  //  64x64
    CFG(2, 4, 4, 1, 1), CFG(2, 4, 4, 1, 2),  // 128x64   (wave 32x64, GEGLU capable)
    CFG(1, 4, 4, 1, 1), CFG(1, 4, 4, 1, 4),  //  64x64   (wave 16x64, GEGLU capable)
    CFG(4, 7, 2, 2, 1), CFG(4, 7, 2, 2, 2),  // 128x224  (7-family: 224 = 7*32)
    CFG(2, 7, 2, 2, 1), CFG(2, 7, 2, 2, 2),  //  64x224
    CFG(1, 7, 2, 2, 1), CFG(1, 7, 2, 2, 2),  //  32x224
    CFG(2, 7, 4, 1, 1), CFG(2, 7, 4, 1, 2),  // 128x112
    CFG(1, 7, 4, 1, 1), CFG(1, 7, 4, 1, 2), CFG(1, 7, 4, 1, 4),  // 64x112
    CFG(1, 7, 2, 1, 1), CFG(1, 7, 2, 1, 4),  //  32x112  (2 waves)
    CFG(2, 1, 4, 1, 1), CFG(2, 1, 4, 1, 4),  // 128x16   (N <= 16: UNet/VAE output convs)
    CFG(1, 2, 4, 1, 1), CFG(1, 2, 4, 1, 4), CFG(1, 2, 4, 1, 8),  // 64x32
    CFG(1, 4, 1, 4, 1), CFG(1, 4, 1, 4, 2),  //  16x256  (tiny M: emb / context projections)
    CFG(1, 2, 2, 2, 1), CFG(1, 2, 2, 2, 4), CFG(1, 2, 2, 2, 8),  // 32x64
    // wave-specialised (4 loader + 4 MFMA waves, LDS-DMA ring of 3 slots)
    CFGW(4, 4, 2, 2, 2, 3), CFGW(2, 4, 2, 2, 2, 3), CFGW(4, 2, 2, 2, 2, 3), CFGW(2, 2, 2, 2, 4, 3),
    CFGW(2, 4, 4, 1, 2, 3), CFGW(1, 4, 4, 1, 4, 3),
    CFGW(4, 7, 2, 2, 2, 3), CFGW(2, 7, 2, 2, 2, 3), CFGW(1, 7, 2, 2, 2, 3),
    CFGW(2, 7, 4, 1, 2, 3), CFGW(1, 7, 4, 1, 4, 3), CFGW(2, 2, 2, 2, 2, 3), CFGW(1, 7, 4, 1, 2, 3),
    // K-split across the MFMA waves (WM = WN = 1): block tile == register tile
    CFGW(2, 7, 1, 1, 4, 3), CFGW(4, 7, 1, 1, 4, 3), CFGW(2, 4, 1, 1, 4, 3), CFGW(4, 4, 1, 1, 4, 3),
    CFGW(2, 2, 1, 1, 4, 3), CFGW(4, 2, 1, 1, 4, 3), CFGW(1, 7, 1, 1, 4, 3), CFGW(4, 8, 1, 1, 4, 3),
    // deep rings (5-7 stages in flight): the 8x8 / 4x4 levels stream 14-29 MB of cold weights per launch with only
    // 64-512 output rows; with 2 stages in flight each workgroup pays one HBM round trip per ~32 KB
    CFGW(2, 4, 2, 2, 2, 6), CFGW(4, 4, 2, 2, 1, 6), CFGW(2, 2, 2, 2, 2, 8), CFGW(2, 7, 4, 1, 1, 6),
    CFGW(1, 4, 4, 1, 2, 8), CFGW(4, 2, 2, 2, 2, 6),
#ifdef UPK_R6_EXPERIMENTS
    // round 6, generation 2 of VERDICT r05 item 1: the 128x224 / 128x128 tiles of the shared-chip table with
    // (a) one-chunk stages on a 6-slot ring (112 KB in flight instead of 90, a barrier per chunk), (b) eight loader waves.
    // Measured by chip time (profiles/r06_gen2_eight_loader_waves_deep_ring_chip_time.txt): 9.72 / 9.90 against 9.67 us on
    // the level-0 3x3 conv — neither the ring depth nor the number of DMA issuers bounds the K loop.  Dev builds only
    // (UPK_CXXFLAGS=-DUPK_R6_EXPERIMENTS): a configuration that does not win does not ship.
    CFGW(4, 7, 2, 2, 1, 6), CFGW8(4, 7, 2, 2, 2, 3), CFGW8(4, 7, 2, 2, 1, 6), CFGW8(4, 4, 2, 2, 2, 3), CFGW8(2, 7, 2, 2, 2, 3),
#endif
    // (256x128 / 128x256 tiles — 85 FLOP per filled byte against 64 — were tried for the VAE decoder's long convs:
    // 128 accumulator registers + the plain / statistics epilogues spill 50-200 VGPRs at 2 waves per SIMD; not kept)
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Rough cycle model used to pick (config, split-K) for shapes that are not in the tuning
// cache (upgpt_amd/tuned_gfx950.json holds measured choices for the bench shapes).
double estimate(const CfgInfo& c, int M, int npad, int nchunks, int splitk, int cus, bool geglu) {
  const int BM = c.mi * 16 * c.wm, BN = c.ni * 16 * c.wn;
  const bool ksplit = c.nbuf && c.wm * c.wn == 1;  // 4 MFMA waves share the tile and split K
  const int waves = ksplit ? 4 : c.wm * c.wn;
  if (geglu && ((c.ni * 16) % 64 != 0)) return 1e30;
  const int tiles = cdiv(M, BM) * cdiv(npad, BN);
  const long wgs = (long)tiles * splitk;
  const int chunks = cdiv(nchunks, splitk);
  const int stages = cdiv(chunks, c.ks);
  // per K-chunk cycles of one workgroup
  const double mfma = c.mi * c.ni * 16.0 * (waves > 4 ? waves / 4.0 : 1.0) / (ksplit ? 4.0 : 1.0);
  const double lds = (c.mi + c.ni) * 4.0 * (ksplit ? 1 : waves) + (c.nbuf ? 0.0 : (BM + BN) * 4 * 13.0 / 64.0 / 4.0);
  const double gl = (BM + BN) * 64.0 / 40.0;  // bytes / (B/clk/CU sustained from L2)
  const double work = fmax(fmax(mfma, lds), gl) * c.ks;
  // workgroups resident per CU (LDS + registers), they overlap each other's stalls
  const int lds_bytes = (c.nbuf ? c.nbuf : 2) * c.ks * (BM + BN) * 64;
  int occ = 160 * 1024 / lds_bytes;
  const int nt = waves * 64;
  const int stage_regs = c.ks * (cdiv(BM * 4, nt) + cdiv(BN * 4, nt)) * 4;
  const int regs = c.mi * c.ni * 4 + (c.mi + c.ni) * 8 + stage_regs + 40;
  int occ_r = (512 / regs) * 4 / waves;
  if (occ_r < 1) occ_r = 1;
  if (occ > occ_r) occ = occ_r;
  if (occ > 4) occ = 4;
  if (occ < 1) occ = 1;
  const double slots = (double)cus * occ;
  const double rounds = ceil(wgs / slots);
  const long resident = wgs < (long)slots ? (wgs + cus - 1) / cus : occ;  // WGs actually sharing a CU
  // one stage: its own work (shared with co-resident WGs) or the exposed load latency
  const double latency = c.nbuf ? 500.0 : 1400.0;  // the DMA ring hides most of the load latency
  const double per_stage = fmax(work * (double)resident, latency) + 150.0;
  double t = rounds * (stages * per_stage + 2500.0);
  if (splitk > 1) t += 6000.0 + (double)M * npad * splitk * 8.0 / (cus * 8.0);
  return t;
}

}  // namespace

// configurations kNumCfgs .. kNumCfgs + astat_num_configs() - 1 are the A-stationary family (astat.hip); for those the
// "split-K" slot of the tuning pair means output-column passes per workgroup (0 = fill the chip once)
// ... and behind those the big-tile family (bigtile.hip); its second slot is a split-K factor like the first families'
extern "C" int upk_conv_num_configs(void) { return kNumCfgs + astat_num_configs() + bt_num_configs(); }
extern "C" const char* upk_conv_config_name(int cfg) {
  if (cfg >= kNumCfgs + astat_num_configs()) return bt_config_name(cfg - kNumCfgs - astat_num_configs());
  if (cfg >= kNumCfgs) return astat_config_name(cfg - kNumCfgs);
  return (cfg >= 0 && cfg < kNumCfgs) ? kCfgs[cfg].name : "?";
}
extern "C" int upk_conv_override(upk_ctx* ctx, int cfg, int splitk) {
  if (!ctx) return UPK_EINVAL;
  if (cfg >= upk_conv_num_configs()) return upk_fail(ctx, UPK_EINVAL, "conv_override: configuration %d of %d", cfg, upk_conv_num_configs());
  ctx->cfg_override = cfg;
  ctx->splitk_override = splitk;
  return UPK_OK;
}

// launch == false: stops after the (config, split-K) decision and reports whether the reduce pass will
// produce GroupNorm partials (upk_conv_gn_fused)
static int conv_impl(upk_ctx* ctx, const upk_conv_desc* d, upk_stream stream_, bool launch, int* gn_fused,
                     int* gn_nblk = nullptr, int* lnr_slots = nullptr) {
  if (lnr_slots) *lnr_slots = 0;
  if (!ctx || !d) return UPK_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  if (!d->x1 || !d->w_packed || !d->y) return upk_fail(ctx, UPK_EINVAL, "conv: null x1/w/y");
  if (d->c1 <= 0 || (d->c1 & 31) || (d->c2 & 31) || d->c2 < 0)
    return upk_fail(ctx, UPK_ESHAPE, "conv: channels must be multiples of 32 (c1=%d c2=%d)", d->c1, d->c2);
  if (d->c2 > 0 && !d->x2) return upk_fail(ctx, UPK_EINVAL, "conv: c2>0 but x2 null");
  if (d->ksize != 1 && d->ksize != 3) return upk_fail(ctx, UPK_ESHAPE, "conv: ksize %d", d->ksize);
  if (d->stride != 1 && d->stride != 2) return upk_fail(ctx, UPK_ESHAPE, "conv: stride %d", d->stride);
  if ((d->ld1 & 7) || (d->c2 && (d->ld2 & 7)) || (d->n_pad & 15) || d->n_pad <= 0)
    return upk_fail(ctx, UPK_EINVAL, "conv: leading dims must be multiples of 8, n_pad of 16");
  const int flags = d->flags;
  const bool geglu = flags & UPK_F_GEGLU;
  if (geglu && (d->n_pad & 63)) return upk_fail(ctx, UPK_ESHAPE, "conv: GEGLU needs n_pad %% 64 == 0");
  if (!(flags & UPK_F_OUT_NCHW_F32) && (d->ldy & 3)) return upk_fail(ctx, UPK_EINVAL, "conv: ldy %% 4");
  if (d->residual && (d->ld_res & 3)) return upk_fail(ctx, UPK_EINVAL, "conv: ld_res %% 4");

  IgemmArgs a;
  memset(&a, 0, sizeof(a));
  a.x1 = (const f16*)d->x1;
  a.x2 = (const f16*)d->x2;
  a.c1 = d->c1;
  a.c2 = d->c2;
  a.ld1 = d->ld1;
  a.ld2 = d->ld2;
  a.w = (const f16*)d->w_packed;
  a.pf = (const char*)d->pf_next;
  a.pf_lines = d->pf_next && d->pf_bytes > 0 ? (int)((d->pf_bytes < (1ll << 37) ? d->pf_bytes : (1ll << 37)) >> 7) : 0;  // (whole lines only)
  a.zero = (const f16*)ctx->zero_page;
  a.npad = d->n_pad;
  a.bias = d->bias;
  a.res = (const f16*)d->residual;
  a.ldr = d->ld_res;
  a.rowvec = d->rowvec;
  a.rv_bs = d->rv_batch_stride;
  a.rv_ss = d->rv_step_stride;
  a.step = d->step;
  a.y = d->y;
  a.ldy = d->ldy;
  a.vt = (f16*)d->vt;
  a.vt_from = d->vt_from;
  a.vt_heads = d->vt_heads;
  a.vt_dhead = d->vt_dhead;
  a.vt_ld = d->vt_ld;
  a.vt_tokens = d->vt_tokens;
  if (a.vt && (a.vt_dhead <= 0 || (a.vt_dhead & 3) || (a.vt_from & 3) || a.vt_tokens <= 0))
    return upk_fail(ctx, UPK_EINVAL, "conv: bad vt_* parameters");
  a.n_out = d->n_out;
  a.B = d->batch;
  a.HS = d->in_h;
  a.WS = d->in_w;
  a.ups = (flags & UPK_F_UPSAMPLE2X) ? 1 : 0;
  a.HL = a.ups ? 2 * a.HS : a.HS;
  a.WL = a.ups ? 2 * a.WS : a.WS;
  a.ks = d->ksize;
  a.stride = d->stride;
  a.ph_on = 0;
  a.ph_wstride = 0;
  static const bool ph_off = getenv("UPK_NO_PHASES") != nullptr;
  const bool phases = d->w_phase && a.ups && d->ksize == 3 && d->stride == 1 && !(flags & UPK_F_PAD_ASYM) && !ph_off &&
                      !d->x3 && !d->ln_colsum && !d->vt && !d->residual && !d->rowvec &&
                      !(flags & (UPK_F_GEGLU | UPK_F_OUT_NCHW_F32));
  const int pad = (d->ksize == 3) ? 1 : 0;
  if (phases) {
    // nearest 2x upsample + conv3x3 = four 2x2 convs on the low-resolution grid (4/9 of the MACs): output pixel
    // (2y + py, 2x + px) sees low-resolution rows {y + py - 1, y + py} and columns {x + px - 1, x + px}
    a.ph_on = 1;
    a.ups = 0;
    a.HL = a.HS;
    a.WL = a.WS;
    a.ks = 2;
    a.pad_lo = 1;  // (the kernels take 1 - py / 1 - px)
    a.Ho = a.HS;
    a.Wo = a.WS;
    a.w = (const f16*)d->w_phase;
  } else if (flags & UPK_F_PAD_ASYM) {
    if (d->stride != 2 || d->ksize != 3) return upk_fail(ctx, UPK_ESHAPE, "conv: PAD_ASYM needs 3x3 s2");
    a.pad_lo = 0;
    a.Ho = (a.HL + 1 - 3) / 2 + 1;
    a.Wo = (a.WL + 1 - 3) / 2 + 1;
  } else {
    a.pad_lo = pad;
    a.Ho = (a.HL + 2 * pad - d->ksize) / d->stride + 1;
    a.Wo = (a.WL + 2 * pad - d->ksize) / d->stride + 1;
  }
  a.M = a.B * a.Ho * a.Wo;
  if (a.M <= 0) return upk_fail(ctx, UPK_EINVAL, "conv: empty output");
  {
    auto lg2 = [](int v) {
      int s = 0;
      while ((1 << s) < v) ++s;
      return (1 << s) == v ? s : -1;
    };
    a.sh_hw = lg2(a.Ho * a.Wo);
    a.sh_w = lg2(a.Wo);
  }
  a.linear = (a.ks == 1 && a.stride == 1 && !a.ups) ? 1 : 0;
  a.ln_u = d->ln_colsum;
  a.ln_eps = d->ln_eps;
  a.ln_inv_dim = d->ln_dim > 0 ? 1.0f / (float)d->ln_dim : 0.f;
  a.lnr_out = nullptr;
  a.lnr_in = a.ln_u ? d->ln_rows_in : nullptr;
  a.lnr_slots = d->ln_rows_slots;
  if (a.lnr_in && (a.lnr_slots < 1 || a.lnr_slots > 8))
    return upk_fail(ctx, UPK_EINVAL, "conv: ln_rows_slots must be 1..8");
  if (a.ln_u && (!a.linear || d->c2 != 0 || d->ln_dim <= 0 || d->ln_dim > d->c1))
    return upk_fail(ctx, UPK_EINVAL, "conv: folded LayerNorm needs a 1x1 stride-1 single-source launch, 0 < ln_dim <= c1");
  a.cpt = (a.c1 + a.c2) / 32;
  a.nchunks_main = a.ks * a.ks * a.cpt;
  a.nchunks = a.nchunks_main;
  if (a.ph_on) a.ph_wstride = a.nchunks_main * a.npad * 32;
  if (d->x3) {
    if (d->c3 <= 0 || (d->c3 & 31) || d->c4 < 0 || (d->c4 & 31) || (d->c4 > 0 && !d->x4) || (d->ld3 & 7) ||
        (d->c4 && (d->ld4 & 7)))
      return upk_fail(ctx, UPK_ESHAPE, "conv: appended segment needs c3, c4 multiples of 32 and ld3, ld4 of 8");
    if (d->stride != 1 || a.ups || (flags & UPK_F_PAD_ASYM) || a.ln_u)
      return upk_fail(ctx, UPK_ESHAPE, "conv: appended 1x1 segment needs stride 1, no upsample, no folded LayerNorm");
    a.x3 = (const f16*)d->x3;
    a.x4 = d->c4 > 0 ? (const f16*)d->x4 : nullptr;
    a.c3 = d->c3;
    a.c4 = d->c4;
    a.ld3 = d->ld3;
    a.ld4 = d->ld4;
    a.nchunks += (a.c3 + a.c4) / 32;
  }
  a.flags = flags;
  if (const char* ab = getenv("UPK_ABLATE")) a.flags |= (int)strtol(ab, nullptr, 0) & 0xFFF0000;
  a.dbg = (unsigned long long*)((char*)ctx->ws + ctx->ws_bytes - 4096);

  // ---- choose config + split-K ----
  int best = -1, best_sk = 1;
  double best_t = 1e30;
  const int sk_cands[] = {1, 2, 3, 4, 6, 8, 9, 12, 16, 18};
  const int nph = a.ph_on ? 4 : 1;
  const size_t slab = (size_t)a.M * a.npad * sizeof(slab_t) * nph;
  const int want_cfg = ctx->cfg_override >= 0 ? ctx->cfg_override : (d->tune_cfg > 0 ? d->tune_cfg - 1 : -1);
  const int want_sk = ctx->splitk_override > 0 ? ctx->splitk_override : (d->tune_splitk > 0 ? d->tune_splitk : 0);
  if (want_cfg >= upk_conv_num_configs())
    return upk_fail(ctx, UPK_ESHAPE, "conv: configuration index %d out of range (%d configurations; stale tuning file?)", want_cfg,
                    upk_conv_num_configs());
  AsPlan aspl;
  const int bt0 = kNumCfgs + astat_num_configs();
  const bool bt_able = !a.x3 && !(a.ln_u && !a.lnr_in);  // (no appended segment, no fragment-side LayerNorm fold)
  bool is_bt = want_cfg >= bt0;
  int bt_bm = 0, bt_bn = 0, bt_occ = 1, bt_mi = 0, bt_ni = 0, bt_wn = 1;
  if (is_bt) {
    bt_tile(want_cfg - bt0, &bt_bm, &bt_bn, &bt_occ, &bt_mi, &bt_ni, &bt_wn);
    const int sk = want_sk > 0 ? want_sk : 1;
    if (!bt_able || (sk == 1 && !Epi::plain(a) && !bt_full_epilogue(want_cfg - bt0)) || (geglu && ((bt_ni * 16) % 64 != 0)) ||
        (a.ln_u && sk > 1) || (sk > 1 && (slab * sk > ctx->ws_bytes || a.nchunks / sk < 4)))
      return upk_fail(ctx, UPK_ESHAPE, "conv: big-tile configuration %s (split-K %d) does not fit this launch",
                      bt_config_name(want_cfg - bt0), sk);
    best = want_cfg;
    best_sk = sk;
  }
  const bool is_as = want_cfg >= kNumCfgs && !is_bt;
  if (is_as) {
    if (!astat_plan(ctx, a, want_cfg - kNumCfgs, want_sk, &aspl) || (want_sk > 1 && want_sk > aspl.npass))
      return upk_fail(ctx, UPK_ESHAPE, "conv: A-stationary configuration %s (passes per workgroup %d) does not fit this launch",
                      astat_config_name(want_cfg - kNumCfgs), want_sk);
    best = want_cfg;
    best_sk = 1;
  }
  for (int c = 0; c < kNumCfgs && !is_as && !is_bt; ++c) {
    if (want_cfg >= 0 && c != want_cfg) continue;
    // folded LayerNorm: row statistics come from the M x N-split wave-specialised kernels, whole K in one block
    if (a.ln_u && !a.lnr_in && !kCfgs[c].fn_ln) continue;
    if (a.x3 && !kCfgs[c].fn_app) continue;  // the appended K segment lives in the wave-specialised loader
    for (int sk : sk_cands) {
      if (want_sk > 0 && sk != want_sk) continue;
      if (a.ln_u && sk > 1) continue;
      if (sk > 1 && (slab * sk > ctx->ws_bytes || a.nchunks / sk < 4)) continue;
      const double t = estimate(kCfgs[c], a.M * nph, a.npad, a.nchunks, sk, ctx->num_cus, geglu);
      if (t < best_t) {
        best_t = t;
        best = c;
        best_sk = sk;
      }
    }
  }
  if (want_cfg < 0 && want_sk <= 1 && bt_able && Epi::plain(a) && best >= 0 && best_sk == 1) {
    // cost model for the big-tile family: MFMA-bound stages (16 cycles per fragment MFMA, four SIMDs in parallel, the
    // workgroups of a CU taking turns) + prologue / epilogue per tile; only where every CU gets a tile
    for (int c = 0; c < bt_num_configs(); ++c) {
      int bm, bn, occ, mi, ni, wn;
      bt_tile(c, &bm, &bn, &occ, &mi, &ni, &wn);
      if (geglu && ((ni * 16) % 64 != 0)) continue;
      const long tiles = (long)cdiv(a.M, bm) * cdiv(a.npad, bn) * nph;
      if (tiles < ctx->num_cus) continue;
      const double rounds = ceil((double)tiles / ((double)ctx->num_cus * occ));
      const double t = rounds * occ * ((double)a.nchunks * (mi * ni * 16.0 * 1.2 + 60.0) + 9000.0);
      if (t < best_t) {
        best_t = t;
        best = bt0 + c;
        best_sk = 1;
        is_bt = true;
        bt_bm = bm, bt_bn = bn, bt_occ = occ, bt_mi = mi, bt_ni = ni, bt_wn = wn;
      }
    }
  }
  if (best < 0) {
    if (want_sk > 1 && slab * want_sk > ctx->ws_bytes)
      return upk_fail(ctx, UPK_EWORKSPACE, "conv: split-K %d needs %zu workspace bytes, have %zu", want_sk,
                      slab * want_sk, ctx->ws_bytes);
    return upk_fail(ctx, UPK_ESHAPE, "conv: no kernel configuration fits (geglu=%d)", (int)geglu);
  }
  const CfgInfo& c = kCfgs[(is_as || is_bt) ? 0 : best];  // (not used by the A-stationary / big-tile families beyond this block)
  const int BM = is_bt ? bt_bm : (is_as ? aspl.bm : c.mi * 16 * c.wm);
  const int BN = is_bt ? bt_bn : (is_as ? aspl.pw * aspl.ppw : c.ni * 16 * c.wn);
  a.tiles_m = cdiv(a.M, BM);
  a.tiles_n = is_as ? aspl.tiles_n : cdiv(a.npad, BN);
  a.chunks_per_split = cdiv(a.nchunks, best_sk);
  const int zdim = cdiv(a.nchunks, a.chunks_per_split);
#ifdef UPK_R6_EXPERIMENTS
  {
    // (dev experiment) cooperative up-front touch of the launch's own weight slices: 0 off, 1 every wave-specialised
    // launch, 2 only where a slice is shared by few M tiles and is large (the 16x16 and deeper levels), 3 = 2 with lanes only
    static const int self_pf = getenv("UPK_SELF_PREFETCH") ? atoi(getenv("UPK_SELF_PREFETCH")) : 0;
    static const int self_tm = getenv("UPK_SELF_PREFETCH_TM") ? atoi(getenv("UPK_SELF_PREFETCH_TM")) : 32;
    const long slice = (long)BN * 64 * a.chunks_per_split;
    a.pf_self = self_pf == 1 || (self_pf >= 2 && a.tiles_m <= self_tm && slice >= (256 << 10));
  }
#endif
  a.partial = (zdim > 1) ? (float*)ctx->ws : nullptr;
  // GroupNorm partials from the reduce pass (see igemm_reduce_gn_kernel)
  // ... or the whole GroupNorm from the reduce pass (igemm_reduce_gnapply_kernel)
  int ga_v = 0, ga_nv = 0;
  if (d->gno_y && !a.ph_on && zdim > 1 && Epi::plain(a) && d->gn_groups > 0 && a.n_out % d->gn_groups == 0 && d->gno_gamma &&
      d->gno_beta) {
    const int cpg = a.n_out / d->gn_groups;
    // one workgroup per (sample, group) pays while a thread holds <= 2 vectors of 4 channels (the 4x4 / 8x8 levels:
    // 5.7 / 8.5 us against 6 + 6.5 us for reduce + GroupNorm launch); measured at 16x16 (2-wide vectors, 7 per thread)
    // 14 us and at 32x32 (scalars, 28 per thread) 36 us — those keep the statistics by-product + apply launch
    const int v = !(cpg & 3) && !(a.ldy & 3) && !(d->gno_ld & 3) && (!a.res || !(a.ldr & 3)) ? 4 : (!(cpg & 1) ? 2 : 1);
    const long nvec = (long)a.Ho * a.Wo * (cpg / v);
    const int nv = (int)((nvec + 255) / 256);
    static const int nv_max = getenv("UPK_GNAPPLY_NVMAX") ? atoi(getenv("UPK_GNAPPLY_NVMAX")) : 2;
    // (never more vectors per thread than the widest instantiation below covers: <4, 8>, <2, 8>, <1, 32>)
    const int nv_inst = v == 1 ? 32 : 8;
    const int nv_lim = v == 4 ? (nv_max < nv_inst ? nv_max : nv_inst) : (nv_max > 2 ? nv_inst : 0);
    if (nv <= nv_lim) ga_v = v, ga_nv = nv;
  }
  const bool gn_apply = ga_v != 0;
  const bool gn_fuse = !gn_apply && !a.ph_on && d->gn_stats_ws && zdim > 1 && Epi::plain(a) && !(a.n_out & 7) && a.n_out <= 2048 &&
                       d->gn_groups > 0 && d->gn_groups <= UPK_GN_GROUPS_MAX && a.n_out % d->gn_groups == 0 &&
                       !(a.ldy & 7) && (!a.res || !(a.ldr & 7));
  // ... or, without split-K, per-(M tile, channel) partials from the plain epilogue (Epi::tile_plain_cp and the
  // K-split kernels' epilogue): needs M tiles that lie inside one sample
  const int hw_out = a.Ho * a.Wo;
  const bool gn_cp = !a.ph_on && d->gn_stats_ws && zdim == 1 && Epi::plain(a) && d->gn_groups > 0 &&
                     d->gn_groups <= UPK_GN_GROUPS_MAX && a.n_out % d->gn_groups == 0 && a.n_out <= 2048 &&
                     hw_out % BM == 0 && hw_out / BM <= (d->gn_stats_cap > UPK_GN_MAX_CHUNKS ? d->gn_stats_cap : UPK_GN_MAX_CHUNKS);
  if (gn_cp) {
    a.gn_cp = d->gn_stats_ws;
    a.gn_nblk = hw_out / BM;
    a.gn_hw = hw_out;
  }
  // LayerNorm row sums of the output for the consumer GEMM (Epi::tile_plain_lnr / the K-split kernels' epilogue)
  if (d->ln_rows_out && !a.ph_on && zdim == 1 && Epi::plain(a) && !gn_cp && (is_as || is_bt || c.wm * c.wn > 1 || c.nbuf > 0)) {
    // (K-split kernels: one slot per N tile; A-stationary: one per 16 * NI columns of its single pass, else none)
    const int slots = is_as ? (aspl.npass == 1 ? cdiv(a.npad, astat_config_ni(best - kNumCfgs) * 16) : 99)
                            : a.tiles_n * (is_bt ? bt_wn : c.wn);
    if (slots <= 8) {
      a.lnr_out = d->ln_rows_out;
      if (lnr_slots) *lnr_slots = slots;
    }
  }
  if (gn_fused) *gn_fused = gn_apply ? 3 : (gn_fuse ? 1 : (gn_cp ? 2 : 0));
  if (gn_nblk) *gn_nblk = gn_cp ? a.gn_nblk : 0;
  if (!launch) return UPK_OK;

#ifdef UPK_TIMELINE
  {  // dev: UPK_TL_TARGET=n stamps only the n-th conv launch of the process (e.g. one launch inside the forward graph)
    static int launch_no = 0;
    static const char* tgt = getenv("UPK_TL_TARGET");
    if (tgt) {
      a.flags &= ~ABL_TIMELINE;
      if (launch_no == atoi(tgt)) {
        a.flags |= ABL_TIMELINE;
        fprintf(stderr, "[timeline] launch %d: M=%d npad=%d nchunks=%d ks=%d cfg=%s(%d) splitk=%d flags=%x res=%d rowvec=%d\n", launch_no,
                a.M, a.npad, a.nchunks, a.ks, upk_conv_config_name(best), best, zdim, flags, a.res != nullptr, a.rowvec != nullptr);
      }
      ++launch_no;
    }
  }
#endif
  upk_prof_scope prof(ctx, UPK_CLS_IGEMM, stream);
  // XCD-aware tile order (tile_map): choose the pm x pn split of the 8 XCDs that minimises the bytes pulled over the
  // fabric — weights by pm XCDs, activations by pn — against the default order (tile index round-robin), measured at
  // 3.5-7x the algorithmic bytes on the 16x16 ... 4x4 levels (scripts/pmc_fetch.sh)
  a.xm_pm = 0;
  a.xm_z = zdim;
  static const int xmap = getenv("UPK_XCD_MAP") ? atoi(getenv("UPK_XCD_MAP")) : 2;  // 0 off, 1 without the 3x3 halo term, 2 default
  dim3 grid(a.tiles_m * a.tiles_n, nph, zdim);
  if (xmap) {
    const long units = (long)a.tiles_n * zdim;
    const double Ab = (double)a.B * a.HS * a.WS * (a.c1 + a.c2) * 2.0 + (double)a.M * (a.c3 + a.c4) * 2.0;
    const double Wb = (double)a.nchunks * 32.0 * a.npad * 2.0;
    auto gcd = [](int x, int y) { while (y) { const int t = x % y; x = y; y = t; } return x; };
    // 3x3 convs: an M tile of r image rows also reads 2 halo rows; scattered over the XCDs every tile fetches its halo
    // itself ((r + 2) / r of the activations), a contiguous run of mi tiles per XCD shares all but its two outer rows
    const bool halo = xmap > 1 && a.ks == 3 && a.stride == 1 && !a.ups && !a.ph_on && BM >= a.Wo;
    const double rows = halo ? (double)BM / a.Wo : 1.0;
    const double h_def = halo ? (rows + 2.0) / rows : 1.0;
    const double cur = Wb * (a.tiles_m < 8 ? a.tiles_m : 8) +
                       Ab * h_def * (double)(units < 8 / gcd(a.tiles_m, 8) ? units : 8 / gcd(a.tiles_m, 8));
    double best = cur * 0.85;  // (the default order unless the gain is clear)
    const int pms[4] = {1, 2, 4, 8};
    for (int pm : pms) {
      const int pn = 8 / pm;
      if (pm > a.tiles_m || pn > units) continue;
      const int mi = cdiv(a.tiles_m, pm), nj = (int)((units + pn - 1) / pn);
      // idle slots = one XCD with less work: harmless while every workgroup has a CU to itself (the grid is smaller
      // than the chip), a longer critical path otherwise (untuned 32x24 forward: +3 % with 20 % slots idle) -> exact fit
      static const double slack_env = getenv("UPK_XCD_SLACK") ? atof(getenv("UPK_XCD_SLACK")) : 0.0;
      const double slack = slack_env > 0.0 ? slack_env : ((long)8 * mi * nj * nph <= ctx->num_cus ? 1.2 : 1.0);
      if ((double)8 * mi * nj > slack * (double)a.tiles_m * units) continue;
      const double c = Wb * pm + Ab * pn * (halo ? (rows * mi + 2.0) / (rows * mi) : 1.0);
      if (c < best) {
        best = c;
        a.xm_pm = pm;
        a.xm_pn = pn;
        a.xm_mi = mi;
        a.xm_nj = nj;
      }
    }
    if (a.xm_pm) grid = dim3(8 * a.xm_mi * a.xm_nj, nph, 1);
  }
  if (is_as) return astat_launch(ctx, a, best - kNumCfgs, aspl, grid, stream);
  int rc;
  if (is_bt) rc = bt_launch(ctx, a, best - bt0, grid, stream);
  else {
  hipLaunchKernelGGL((a.ln_u && !a.lnr_in) ? c.fn_ln : (a.x3 ? c.fn_app : c.fn), grid, dim3(c.nbuf ? 256 + 64 * c.lw : c.wm * c.wn * 64), 0, stream, a);
  rc = upk_check_launch(ctx, "igemm");
  }
  if (rc) return rc;
  if (zdim > 1 && gn_apply) {
    GnApply g;
    g.gamma = d->gno_gamma;
    g.beta = d->gno_beta;
    g.yn = (f16*)d->gno_y;
    g.eps = d->gno_eps;
    g.ldn = d->gno_ld;
    g.silu = d->gno_silu;
    g.skip_y = d->gno_skip_y;
    g.cpg = a.n_out / d->gn_groups;
    g.hw = a.Ho * a.Wo;
    const dim3 gg(d->gn_groups, a.B);
    if (ga_v == 4 && ga_nv <= 1)
      hipLaunchKernelGGL((igemm_reduce_gnapply_kernel<4, 1>), gg, dim3(256), 0, stream, a, zdim, g);
    else if (ga_v == 4 && ga_nv <= 2)
      hipLaunchKernelGGL((igemm_reduce_gnapply_kernel<4, 2>), gg, dim3(256), 0, stream, a, zdim, g);
    else if (ga_v == 4)
      hipLaunchKernelGGL((igemm_reduce_gnapply_kernel<4, 8>), gg, dim3(256), 0, stream, a, zdim, g);
    else if (ga_v == 2)
      hipLaunchKernelGGL((igemm_reduce_gnapply_kernel<2, 8>), gg, dim3(256), 0, stream, a, zdim, g);
    else
      hipLaunchKernelGGL((igemm_reduce_gnapply_kernel<1, 32>), gg, dim3(256), 0, stream, a, zdim, g);
    rc = upk_check_launch(ctx, "igemm_reduce_gnapply");
  } else if (zdim > 1 && gn_fuse) {
    GnFuse gf;
    gf.ws = d->gn_stats_ws;
    gf.groups = d->gn_groups;
    gf.cpg = a.n_out / d->gn_groups;
    gf.hw = a.Ho * a.Wo;
    upk_gn_chunking(gf.hw, &gf.nchunks, &gf.pix_per_chunk);
    hipLaunchKernelGGL(igemm_reduce_gn_kernel, dim3(gf.nchunks, a.B), dim3(256), 0, stream, a, zdim, gf);
    rc = upk_check_launch(ctx, "igemm_reduce_gn");
  } else if (zdim > 1) {
    const long total = (long)a.M * (a.npad / 4);
    hipLaunchKernelGGL(igemm_reduce_kernel, dim3((unsigned)((total + 255) / 256), nph), dim3(256), 0, stream, a, zdim);
    rc = upk_check_launch(ctx, "igemm_reduce");
  }
  return rc;
}

extern "C" int upk_conv2d_nhwc_f16(upk_ctx* ctx, const upk_conv_desc* d, upk_stream stream) {
  return conv_impl(ctx, d, stream, true, nullptr);
}

extern "C" int upk_conv_ln_rows(upk_ctx* ctx, const upk_conv_desc* d, int* slots) {
  if (!slots) return UPK_EINVAL;
  return conv_impl(ctx, d, nullptr, false, nullptr, nullptr, slots);
}

extern "C" int upk_conv_gn_fused(upk_ctx* ctx, const upk_conv_desc* d, int* mode, int* nblk) {
  if (!mode || !nblk) return UPK_EINVAL;
  *mode = 0;
  *nblk = 0;
  return conv_impl(ctx, d, nullptr, false, mode, nblk);
}

constexpr float kGnStatsLaunchUs = 5.0f;  // gn_stats_kernel inside the replayed forward (rocprof: 5.7 us average)

extern "C" int upk_conv_autotune(upk_ctx* ctx, const upk_conv_desc* d, upk_stream stream_, int reps, int* best_cfg,
                                 int* best_splitk, float* best_us, float* default_us) {
  if (!ctx || !d || !best_cfg || !best_splitk) return UPK_EINVAL;
  if (ctx->prof_on) return upk_fail(ctx, UPK_EINVAL, "autotune with profiling enabled");
  hipStream_t stream = (hipStream_t)stream_;
  if (reps < 1) reps = 1;
  // UPK_TUNE_COLD=1: every timed launch runs behind a 512 MB memset, i.e. with the L2s and the
  // Infinity Cache flushed — inside the UNet forward a launch never finds its weights (850 MB cycle
  // through per forward) or its freshly produced activations in the local L2, which back-to-back
  // launches of one shape do.  The flush buffer belongs to the context (freed by upk_destroy).
  static const bool cold = getenv("UPK_TUNE_COLD") != nullptr;
  const size_t flush_bytes = (size_t)512 << 20;
  if (cold && !ctx->tune_flush) UPK_HIP(ctx, hipMalloc(&ctx->tune_flush, flush_bytes));
  void* const flush_buf = ctx->tune_flush;
  hipEvent_t e0, e1;
  UPK_HIP(ctx, hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) {
    (void)hipEventDestroy(e0);
    return upk_fail(ctx, UPK_EHIP, "autotune: hipEventCreate failed");
  }
  const int save_cfg = ctx->cfg_override, save_sk = ctx->splitk_override;
  upk_conv_desc dd = *d;
  dd.tune_cfg = 0;
  dd.tune_splitk = 0;
  auto time_one = [&](int cfg, int sk, float* us) -> int {
    ctx->cfg_override = cfg;
    ctx->splitk_override = sk;
    int rc = upk_conv2d_nhwc_f16(ctx, &dd, stream);  // warm-up + feasibility
    if (rc) return rc;
    if (cold) {
      float tot = 0.f;
      for (int r = 0; r < reps; ++r) {
        if (hipMemsetAsync(flush_buf, r, flush_bytes, stream) != hipSuccess) return UPK_EHIP;
        if (hipEventRecord(e0, stream) != hipSuccess) return UPK_EHIP;
        rc |= upk_conv2d_nhwc_f16(ctx, &dd, stream);
        if (hipEventRecord(e1, stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return UPK_EHIP;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return UPK_EHIP;
        tot += ms;
      }
      *us = tot * 1000.f / reps;
      return rc;
    }
    if (hipEventRecord(e0, stream) != hipSuccess) return UPK_EHIP;
    for (int r = 0; r < reps; ++r) rc |= upk_conv2d_nhwc_f16(ctx, &dd, stream);
    if (hipEventRecord(e1, stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return UPK_EHIP;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return UPK_EHIP;
    *us = ms * 1000.f / reps;
    return rc;
  };
  float dflt = 0.f;
  int rc = time_one(-1, 0, &dflt);
  float best = 1e30f;
  int bc = -1, bs = 1;
  if (rc == UPK_OK) {
    const int sks[] = {1, 2, 3, 4, 6, 8, 9, 12, 16, 18};
    for (int c = 0; c < upk_conv_num_configs(); ++c)
      for (int sk : sks) {
        float us = 0.f;
        if (time_one(c, sk, &us) != UPK_OK) continue;  // infeasible candidate
        // a split launch whose reduce pass writes the GroupNorm partials saves the consumer's gn_stats launch
        int fused = 0;
        if (dd.gn_stats_ws && conv_impl(ctx, &dd, nullptr, false, &fused) == UPK_OK && fused) us -= kGnStatsLaunchUs;  // either mode
        if (us < best) {
          best = us;
          bc = c;
          bs = sk;
        }
      }
    ctx->err[0] = 0;
  }
  ctx->cfg_override = save_cfg;
  ctx->splitk_override = save_sk;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc) return rc;
  if (bc < 0) return upk_fail(ctx, UPK_ESHAPE, "autotune: no feasible configuration");
  *best_cfg = bc;
  *best_splitk = bs;
  if (best_us) *best_us = best;
  if (default_us) *default_us = dflt;
  return UPK_OK;
}

extern "C" int upk_gemm_f16(upk_ctx* ctx, const void* A, int lda, int m, int k, const void* w_packed,
                            int n_out, int n_pad, const float* bias, const void* residual, int ld_res,
                            void* y, int ldy, int flags, upk_stream stream) {
  upk_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.x1 = A;
  d.c1 = k;
  d.ld1 = lda;
  d.batch = 1;
  d.in_h = m;
  d.in_w = 1;
  d.ksize = 1;
  d.stride = 1;
  d.w_packed = w_packed;
  d.n_out = n_out;
  d.n_pad = n_pad;
  d.bias = bias;
  d.residual = residual;
  d.ld_res = ld_res;
  d.y = y;
  d.ldy = ldy;
  d.flags = flags;
  return upk_conv2d_nhwc_f16(ctx, &d, stream);
}

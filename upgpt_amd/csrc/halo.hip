// Halo-patch 3x3 convolution family of upk_conv2d_nhwc_f16 (configurations "hc<NI>p<PF>", behind the big-tile ones).
//
// Why another family (profiles/r04_timeline_conv224.txt, scripts/ubench/dual.hip): the wave-specialised implicit-GEMM
// kernel runs the UNet's 3x3 convs with one 64 x 112 / 64 x 64 tile per CU, both operands through the LDS-DMA ring:
// 709 KB per workgroup at 29 B/clk = 24.5 k of its 43.7 k cycles, the im2col rows fetched nine times in 64-byte pieces.
// A CU pulls a weight slice it touches once at 40-46 B/clk when all eight waves stream it straight into registers
// (16 KB in flight per wave), and the activations a tile needs are only its halo patch.  So here
//   * the input patch of the tile's 64 output pixels — (rows + 2) x (W + 2) pixels, zero padding included — is staged in
//     LDS once per channel range (LDS-DMA, `cr` 32-channel chunks per round, two slots), and the nine taps read it at
//     nine pixel offsets: one fill per input pixel instead of nine;
//   * the weights never touch the LDS: every wave streams ITS K items' fragments (NI x 1 KiB per item: in the packed
//     layout [K/32][n_pad][32] the NI * 16 columns of a tile are one contiguous run) into a PF-deep register ring;
//   * the eight waves split K (item q of the workgroup's (channel chunk, tap) list goes to wave q % 8), each holding
//     the whole 64 x (NI * 16) accumulator tile; the eight partial tiles are summed through LDS at the end and every
//     wave finishes its share of the fragments — the epilogue (operand loads, stores, GroupNorm partials) runs on all
//     eight waves instead of four;
//   * no workgroup barrier inside a round; one barrier per round boundary (all patch DMAs of the next round landed, all
//     waves done with the slot that is refilled next).
//
// LDS layout of a patch chunk: [16-pixel group][pixel][4 x 16-byte pieces], piece position = piece ^ 2 * (pixel / 4 & 1).
// A fragment read at tap (ky, kx) takes 16 CONSECUTIVE patch pixels from an arbitrary start; with this position rule
// the four lanes of a ds_read_b128 service group that share (pixel mod 4) sit 4, 8 and 12 pixels apart and land on
// four different piece positions for every start — conflict-free at every tap (the igemm ring's XOR rule is
// conflict-free only for 16-aligned starts).
//
// Scope: ksize 3, stride 1, pad 1, no upsample, W in {4, 8, 16, 32, 64} with 64 output pixels = whole image rows or
// whole images; one or two sources plus the appended 1x1 segment (a ResBlock's skip projection: taps = centre only);
// split-K over channel ranges with the usual partial slabs (the reduce passes of igemm.hip follow); plain epilogue (bias,
// timestep row vector, residual -> fp16 NHWC, GroupNorm channel partials as a by-product) or slabs.
//
// Input GroupNorm (template flag GNI; include/upk.h gni_*): a ResBlock runs GroupNorm32 -> SiLU -> conv3x3 twice
// (openaimodel.py:203-206, 227-233).  The normalised tensor exists only to be read by the conv, and the patch is the one
// place where every input pixel passes exactly once per workgroup — so with GNI the 3x3 sources x1 | x2 are the
// UN-normalised tensors: the patch goes global -> registers -> (x * scale + shift, SiLU) -> LDS instead of through the
// LDS-DMA (a DMA cannot transform data), in the DMA's LDS layout; padding pixels stay exact zeros.  Every workgroup first
// folds its sample's per-(row block, channel) partial sums (the producers' epilogue by-products) into the per-channel
// scale / shift table — gn_apply_kernel's arithmetic (norm.hip), step for step, so the conv's result is bit-identical to
// that launch followed by this kernel without GNI.  The raw patch loads, the weight ring's first loads and the partial
// sums are requested together: one memory round trip, then LDS-only work.  The `gn_apply` launch (5-7 us at the 32x32 and
// 16x16 levels, 13-20 us for the concatenated inputs) and its 2 x tensor bytes disappear.
#include "igemm_common.h"

namespace upkd {
namespace {

constexpr int HC_NW = 8;
constexpr int HC_GNI_K = 16;  // GNI: (chunk, 16-pixel group) units of the patch per wave = raw 16-byte vectors per lane
constexpr int HC_PFN = 4;  // prefetch requests per wave (items rank + k * G): 4 x G items cover the 63-item slice of the 32x32 level

struct HcArgs {
  const f16* x1;
  const f16* x2;
  const f16* x3;
  const f16* x4;
  const f16* w;
  const f16* zero;
  float* partial;
  int ch1, ch2, ch3, ch4;  // 32-channel chunks per source (x1 | x2: the 3x3 taps, x3 | x4: appended 1x1)
  int ld1, ld2, ld3, ld4;
  int npad, M, H, W;
  int pw;         // patch row pitch in pixels: W + 2
  int part_pix;   // patch pixels per part: (rows_part + 2) * pw
  int hw;         // H * W
  int npix;       // patch pixels of the tile
  int ngrp;       // 16-pixel groups of the patch (1 KiB per group and chunk)
  int cr;         // chunks per round
  int nslot;      // patch slots in LDS (1: the whole K range of a workgroup fits one round)
  int cpt;        // chunks per tap: ch1 + ch2
  int mps, aps;   // 3x3 / appended chunks per K split
  int cp_off;     // byte offset of the GroupNorm-partials scratch behind ring / reduction buffer
  int tab_off;    // byte offset of the K-item table behind it
  int pp_magic, pw_magic;  // ceil(65536 / part_pix), ceil(65536 / pw): exact quotients of patch pixel indices (< 256)
  int sh_hw, sh_w;  // log2 of H * W / W (powers of two by construction)
  int gni_off;      // GNI: byte offset of the [2 C] channel sums -> scale | shift table (+ group scratch) behind the item table
  int ng_magic;     // GNI: ceil(65536 / ngrp)
};
#define HC_PIN(v) asm volatile("" ::"s"(v))

// global -> LDS, 16 bytes per lane, LDS address = m0 + lane * 16.  Inline asm on purpose: to the compiler's wait-count
// model a pending LDS-DMA turns every later wait it inserts into vmcnt(0) lgkmcnt(0), which would
// drain the weight ring in front of every K item while the next round's patch is in flight.  Unknown to the model,
// the DMAs only make its counted waits conservative (the hardware counter is in order).  m0 is not used by anything
// else in this kernel.
__device__ __forceinline__ void hc_dma16(const void* src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory");
}

// one round of the workgroup's K range: `n` channel chunks of ONE source starting at global chunk `g`
// (global numbering: [0, cpt) = the 3x3 sources x1 | x2, [cpt, cpt + ch3 + ch4) = the appended sources x3 | x4)
struct HcRound {
  int g, n;
  int idx;   // round number
  int base;  // first LDS chunk position of the round: (idx & 1) * cr with two slots, chunks before it with one
};

// MI: 16-pixel fragments per tile (4: 64 output pixels, 8: 128 — twice the pixels per streamed weight byte, which is what
// bounds the K loop: DESIGN.md 11b)
template <int MI, int NI, int PF, bool GNI>
__global__ __launch_bounds__(512) void halo_conv_kernel(const HcArgs s, const IgemmArgs a) {
  constexpr int NW = HC_NW, NF = MI * NI, BM = MI * 16;
  constexpr int KS = 4;             // K slices: wave = (N half) * 4 + (K slice)
  constexpr int NJ = (NI + 1) / 2;  // column fragments per wave (the second half of an odd NI carries one dead fragment)
  constexpr int FW = MI * NJ;       // accumulator fragments per wave
  static_assert(FW == KS * NJ, "the epilogue finishes NJ = FW / KS fragments per wave: 64-pixel tiles");
  extern __shared__ __attribute__((aligned(16))) f16 smem[];

  HC_PIN(s.x1); HC_PIN(s.x2); HC_PIN(s.x3); HC_PIN(s.x4); HC_PIN(s.w); HC_PIN(s.zero); HC_PIN(s.partial);
  HC_PIN(s.ch1); HC_PIN(s.ch2); HC_PIN(s.ch3); HC_PIN(s.ch4); HC_PIN(s.ld1); HC_PIN(s.ld2); HC_PIN(s.ld3); HC_PIN(s.ld4);
  HC_PIN(s.npad); HC_PIN(s.M); HC_PIN(s.H); HC_PIN(s.W); HC_PIN(s.pw); HC_PIN(s.part_pix); HC_PIN(s.hw); HC_PIN(s.npix);
  HC_PIN(s.ngrp); HC_PIN(s.cr); HC_PIN(s.nslot); HC_PIN(s.cpt); HC_PIN(s.mps); HC_PIN(s.aps); HC_PIN(s.cp_off);
  HC_PIN(s.sh_hw); HC_PIN(s.sh_w); HC_PIN(s.tab_off); HC_PIN(s.pp_magic); HC_PIN(s.pw_magic);
  // the timestep row of the epilogue's row vector: a dependent scalar load in front of the operand loads otherwise
  const int st0 = (a.rowvec && a.step) ? *a.step : 0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = wave & (KS - 1), nh = wave >> 2;
  const int lg = lane >> 4, lc = lane & 15;
#ifdef UPK_TIMELINE
  const bool tl = (a.flags & ABL_TIMELINE) && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && blockIdx.z == 0 && wave == 0;
  unsigned long long* tlp = a.dbg + (blockIdx.x == 0 ? 0 : 32);
#define STAMP(i) do { if (tl && lane == 0) tlp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
  STAMP(0);
  int tm, tn, zs;
  if (!tile_map(a, tm, tn, zs)) return;
  const int m0 = tm * BM;
  const int n0 = tn * (NI * 16);

  // ---- K range of this split: 3x3 chunks [mlo, mhi) and appended chunks [alo, ahi), global numbering
  const int cpt = s.cpt, capp = s.ch3 + s.ch4;
  const int mlo = min(cpt, zs * s.mps), mhi = min(cpt, mlo + s.mps);
  const int alo = cpt + min(capp, zs * s.aps), ahi = min(cpt + capp, alo + s.aps);
  const int b1 = s.ch1, b3 = cpt + s.ch3;
  const int kEnd = 0x7fffffff;
  // the round that starts at global chunk g (g == kEnd: none): up to cr chunks, never across a source or range boundary
  auto round_at = [&](int g, int idx, int cum) -> HcRound {
    HcRound r;
    r.g = g;
    int lim = g < b1 ? b1 : (g < cpt ? cpt : (g < b3 ? b3 : cpt + capp));
    lim = min(lim, g < cpt ? mhi : ahi);
    r.n = g == kEnd ? 0 : min(s.cr, lim - g);
    r.idx = idx;
    r.base = s.nslot > 1 ? (idx & 1) * s.cr : cum;
    return r;
  };
  auto next_round = [&](const HcRound& r) -> HcRound {
    int g = kEnd;
    if (r.g != kEnd) {
      const int e = r.g + r.n;
      if (e <= cpt && r.g < cpt) g = e < mhi ? e : (alo < ahi ? alo : kEnd);
      else g = e < ahi ? e : kEnd;
    }
    return round_at(g, r.idx + 1, s.nslot > 1 ? 0 : r.base + r.n);
  };
  const int g_first = mlo < mhi ? mlo : (alo < ahi ? alo : kEnd);
  const HcRound r_first = round_at(g_first, 0, 0);

  // ---- patch geometry of this lane
  // (a) as a DMA lane: patch pixel P = group * 16 + lane / 4, piece (lane & 3) swizzled.  Wave w owns group w in every
  // chunk; the groups >= 8 (the patch of a 32-wide level has 9, of a 64-wide one 13) are shared out chunk by chunk
  const int b0 = m0 >> s.sh_hw;                          // first image of the tile
  const int y0 = (m0 - (b0 << s.sh_hw)) >> s.sh_w;       // first output row inside it (0 when the tile holds whole images)
  const int piece = (lane & 3) ^ (((lane >> 4) & 1) << 1);
  const int nextra = max(0, s.ngrp - 8);
  const int xg = nextra > 0 ? 8 + wave % nextra : -1;                       // this wave's extra group ...
  const int xrank = nextra > 0 ? wave / nextra : 0;                         // ... of which it takes chunks xrank, xrank + xshare, ...
  const int xshare = nextra > 0 ? (8 - wave % nextra + nextra - 1) / nextra : 1;
  int pix[2];   // linear input pixel (b * H + iy) * W + ix of the lane's patch pixel, or -1 (padding / past the patch)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int grp = q == 0 ? wave : xg;
    const int P = grp * 16 + (lane >> 2);
    const int part = (P * s.pp_magic) >> 16, rem = P - part * s.part_pix;
    const int py = (rem * s.pw_magic) >> 16, px = rem - py * s.pw;
    const int iy = y0 + py - 1, ix = px - 1;
    const int b = b0 + part;
    const bool ok = grp >= 0 && grp < s.ngrp && P < s.npix && iy >= 0 && iy < s.H && ix >= 0 && ix < s.W && (b << s.sh_hw) < s.M;
    pix[q] = ok ? ((b * s.H + iy) << s.sh_w) + ix : -1;
  }
  const f16* zsrc = s.zero + (lane & 3) * 8;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  // issues this wave's DMAs of round r
  auto issue_round = [&](const HcRound& r) {
    if (r.n <= 0) return;
    const int g = r.g;
    const bool app = g >= cpt;
    const f16* sp = g < b1 ? s.x1 : (!app ? s.x2 : (g < b3 ? s.x3 : s.x4));
    const int sld = g < b1 ? s.ld1 : (!app ? s.ld2 : (g < b3 ? s.ld3 : s.ld4));
    const int c0 = (g - (g < b1 ? 0 : (!app ? b1 : (g < b3 ? cpt : b3)))) * 32;  // first channel inside the source
    const unsigned dst0 = lds0 + (unsigned)(r.base * s.ngrp) * 1024u;
    if (wave < s.ngrp) {
      const f16* src = pix[0] >= 0 ? sp + (long)pix[0] * sld + c0 + piece * 8 : zsrc;
      const int step = pix[0] >= 0 ? 32 : 0;
      for (int cl = 0; cl < r.n; ++cl) hc_dma16(src + cl * step, dst0 + (unsigned)((cl * s.ngrp + wave) * 1024));
    }
    if (xg >= 0) {
      const f16* src = pix[1] >= 0 ? sp + (long)pix[1] * sld + c0 + piece * 8 : zsrc;
      const int step = pix[1] >= 0 ? 32 : 0;
      for (int cl = xrank; cl < r.n; cl += xshare) hc_dma16(src + cl * step, dst0 + (unsigned)((cl * s.ngrp + xg) * 1024));
    }
  };
  // (b) as an MFMA lane: patch pixel of tile pixel t = i * 16 + lc at tap (0, 0)
  unsigned Pl[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int t = i * 16 + lc;
    const int part = t >> s.sh_hw;  // (0 unless the tile holds whole images: H * W < BM)
    const int rem = t - (part << s.sh_hw);
    const int ry = rem >> s.sh_w, x = rem - (ry << s.sh_w);
    Pl[i] = (unsigned)(part * s.part_pix + ry * s.pw + x);
  }
  const unsigned lg16 = (unsigned)lg * 16u;

  // ---- the patch on its way: every round with one slot (no position is reused), rounds 0 and 1 with two
  HcRound dr = r_first;  // DMA walker: the next round to issue
  int total_rounds = 0;  // rounds of this workgroup = availability steps (barriers) every wave takes
  {
    HcRound t = r_first;
    while (t.n > 0) {
      ++total_rounds;
      t = next_round(t);
    }
  }
  for (int k = 0; k < (s.nslot > 1 ? 2 : total_rounds); ++k) {
    if (!(GNI && dr.g < cpt)) issue_round(dr);  // (GNI: the 3x3 sources go through registers, below)
    dr = next_round(dr);
  }
  // ---- GNI: the raw patch of the 3x3 sources on its way into registers.  One slot (hc_plan), so main chunk g sits at
  // LDS chunk position g - mlo; unit u = wave + 8 k of the (chunk, 16-pixel group) list is this wave's k-th: chunk
  // u / ngrp, group u % ngrp, lane = (pixel, piece) exactly as the DMA would place it
  constexpr int GK = GNI ? HC_GNI_K : 1;
  f16x8 raw[GK];
  unsigned gni_ok = 0;  // bit k: raw[k] is a real pixel (else padding / past the patch: stays zero)
  const int gni_units = GNI ? (mhi - mlo) * s.ngrp : 0;
  if (GNI) {
#pragma unroll
    for (int k = 0; k < GK; ++k) {
      const int u = wave + HC_NW * k;
      raw[k] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (u < gni_units) {
        const int cl = (u * s.ng_magic) >> 16, grp = u - cl * s.ngrp;
        const int P = grp * 16 + (lane >> 2);
        const int part = (P * s.pp_magic) >> 16, rem = P - part * s.part_pix;
        const int py = (rem * s.pw_magic) >> 16, px = rem - py * s.pw;
        const int iy = y0 + py - 1, ix = px - 1;
        const bool ok = P < s.npix && part == 0 && iy >= 0 && iy < s.H && ix >= 0 && ix < s.W;
        const int g = mlo + cl;
        const bool second = g >= b1;
        const f16* sp = second ? s.x2 : s.x1;
        const int sld = second ? s.ld2 : s.ld1;
        const int c0 = (g - (second ? b1 : 0)) * 32 + piece * 8;
        if (ok) {
          raw[k] = *(const f16x8*)(sp + (long)(((b0 * s.H + iy) << s.sh_w) + ix) * sld + c0);
          gni_ok |= 1u << k;
        }
      }
    }
  }
  STAMP(1);

  // ---- the item table: the workgroup's K items in order (round by round; inside a round tap major, channel chunk
  // minor = the order of the packed weight, so that the workgroups of a launch walk it front to back), one 8-byte entry
  // each — x = byte offset of the item's weight chunk, y = LDS KiB offset of its patch chunk | tap pixel offset << 8 |
  // round << 16 (round 0xffff: no item; 4 * (PF + 2) of those close the table).  Walking the rounds per item in the K
  // loop cost ~250 scalar instructions per 16 MFMAs: a wave issues one instruction per ~4.5 cycles.
  const int kDead = 0xffff;
  unsigned* const tab = (unsigned*)((char*)smem + s.tab_off);
  const unsigned kstr = (unsigned)s.npad * 64u;  // bytes per K chunk of the packed weight
  auto entry = [&](const HcRound& r, int loc, unsigned& ex, unsigned& ey) {
    const bool app = r.g >= cpt;
    const int tap = app ? 4 : (int)(((float)loc + 0.5f) * __builtin_amdgcn_rcpf((float)r.n));
    const int cl = app ? loc : loc - tap * r.n;
    const int ky = (tap * 43) >> 7, kx = tap - 3 * ky;
    const int kc = app ? 9 * cpt + (r.g - cpt) + cl : tap * cpt + r.g + cl;
    ex = (unsigned)kc * kstr;
    ey = (unsigned)((r.base + cl) * s.ngrp) | ((unsigned)(ky * s.pw + kx) << 8) | ((unsigned)r.idx << 16);
  };
  // ---- weight ring: slot u holds the NJ fragments of the item PF steps ahead of the one being multiplied; the item's
  // A-side coordinates (entry.y) travel with the slot
  const int jbase = nh * NJ;  // first column fragment of this wave
  const char* const wb = (const char*)s.w + (size_t)n0 * 64u;
  unsigned vo[NJ];  // per-lane byte offset of fragment j: element (row lc, k group lg) of its [16][32] fp16 run (a dead
                    // fragment — past NI or n_pad — re-reads a live one: one instruction stream, products never stored)
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int jj = jbase + j;
    const bool live = jj < NI && n0 + jj * 16 < s.npad;
    vo[j] = (unsigned)(lc * 64 + lg * 16) + (live ? (unsigned)jj * 1024u : (n0 + jbase * 16 < s.npad ? (unsigned)jbase * 1024u : 0u));
  }
  f16x8 ring[PF][NJ];
  unsigned sI[PF];
  unsigned last_off = 0;  // weight offset of the last live item requested (items past the end re-request it: a cache hit)
  // the first PF items of this wave straight from the round structure when round 0 holds them all (the usual case):
  // their weights are on the way before the table exists
  const bool direct = r_first.n * (r_first.g >= cpt ? 1 : 9) >= 4 * PF;
  if (direct) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      unsigned ex, ey;
      entry(r_first, ks + 4 * u, ex, ey);
      ex = __builtin_amdgcn_readfirstlane(ex), ey = __builtin_amdgcn_readfirstlane(ey);
      sI[u] = ey;
      last_off = ex;
      const char* sb = wb + ex;
#pragma unroll
      for (int j = 0; j < NJ; ++j) ring[u][j] = *(const f16x8*)(sb + vo[j]);
    }
  }
  int nitems = 0;
  {
    int first = 0;
    for (HcRound r = r_first; r.n > 0; r = next_round(r)) {
      const int cnt = r.n * (r.g >= cpt ? 1 : 9);
      for (int loc = tid; loc < cnt; loc += 512) {
        unsigned ex, ey;
        entry(r, loc, ex, ey);
        tab[2 * (first + loc)] = ex;
        tab[2 * (first + loc) + 1] = ey;
      }
      first += cnt;
    }
    if (tid < 4 * (PF + 2)) {
      tab[2 * (first + tid)] = 0u;
      tab[2 * (first + tid) + 1] = (unsigned)kDead << 16;
    }
    nitems = first;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // ---- cooperative prefetch of the weight slice into the XCD's L2.  The workgroups that share this (N tile, K split)
  // on one XCD (tile_map: xm_mi consecutive M tiles; default order: every 8th) run in lockstep and ask for the same
  // lines at the same time, so each of them waits out the full HBM latency for every line: 64 KB in flight per CU over
  // ~2 k cycles = 20-25 B/clk.  Here every one of them first requests ITS share of the slice (items rank, rank + G,
  // ...: one fragment per wave, results dropped): the whole slice is on its way from HBM at once, and the ring's
  // requests find it in L2.  Placement (workgroup w on XCD w % 8) is assumed for speed only.
  f32x4 pfx[HC_PFN];
  {
    const int G = a.xm_pm ? a.xm_mi : max(1, a.tiles_m >> 3);
    const int rank = a.xm_pm ? (int)((blockIdx.x >> 3) % (unsigned)a.xm_mi) : (tm >> 3);
    const unsigned pvo = (unsigned)(min(wave, NI - 1) * 1024 + lane * 16);
#pragma unroll
    for (int k = 0; k < HC_PFN; ++k) {
      const int it = min(rank + k * G, nitems - 1);
      const unsigned off = __builtin_amdgcn_readfirstlane(tab[2 * it]);
      pfx[k] = *(const f32x4*)(wb + off + pvo);
    }
  }
  const uint2* const tq = (const uint2*)tab + ks;  // this wave's entries: tq[4 k]
  if (!direct) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const uint2 e = tq[4 * u];
      const unsigned ex = __builtin_amdgcn_readfirstlane(e.x), ey = __builtin_amdgcn_readfirstlane(e.y);
      sI[u] = ey;
      if ((ey >> 16) != (unsigned)kDead) last_off = ex;
      const char* sb = wb + last_off;
#pragma unroll
      for (int j = 0; j < NJ; ++j) ring[u][j] = *(const f16x8*)(sb + vo[j]);
    }
  }
  // ---- operands of the fragments this wave will finish (bias, timestep row, residual): requested now, one memory round
  // trip under the K loop instead of one in front of the stores
  const Epi::Plain P(a, st0);
  slab_t* slab = s.partial ? (slab_t*)s.partial + (long)zs * a.M * a.npad : nullptr;
  Epi::Plain::Row rows[NJ];
  f32x4 bvs[NJ], rvs[NJ];
  f16x4 rrs[NJ];
#pragma unroll
  for (int q = 0; q < NJ; ++q) {
    const int fw = ks + KS * q;
    const int i = fw / NJ, jj = jbase + (fw - i * NJ);
    const int n = n0 + jj * 16 + lg * 4;
    rows[q] = P.row(a, jj < NI ? m0 + i * 16 + lc : a.M, a.M);
    bvs[q] = P.bias4(a, n);
    rvs[q] = P.rv4(a, rows[q], n);
    rrs[q] = P.res4(a, rows[q], n);
  }
  uint2 enext = tq[4 * PF];  // entry of the next refill, read one step ahead of its use
  int qn = 4 * (PF + 1);
  if (GNI) {
    // ---- scale / shift table of this tile's sample: gn_apply_kernel's fold (norm.hip), the same operations in the
    // same order — blocks per channel in fp32, channels per group in fp64, scale = rstd * gamma, shift = beta - mean * scale
    float* const gt = (float*)((char*)smem + s.gni_off);  // [2][C] channel sums, then scale | shift
    double* const gsum = (double*)(gt + 2 * cpt * 32);      // [groups][2]
    float* const smean = (float*)(gsum + 2 * UPK_GN_GROUPS_MAX);
    float* const srstd = smean + UPK_GN_GROUPS_MAX;
    const int Cn = cpt * 32, c1n = s.ch1 * 32;
    const int cpg = Cn / a.gni_groups;
    float pg[2], pb[2];  // (C <= 1024: two channels per thread)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int ch = tid + k * 512;
      pg[k] = ch < Cn ? a.gni_gamma[ch] : 0.f;
      pb[k] = ch < Cn ? a.gni_beta[ch] : 0.f;
    }
    const float* w1 = a.gni_st1 + (long)b0 * a.gni_nblk1 * 2 * a.gni_ld1;
    const float* w2 = a.gni_st2 ? a.gni_st2 + (long)b0 * a.gni_nblk2 * 2 * a.gni_ld2 : nullptr;
    for (int idx = tid; idx < 2 * Cn; idx += 512) {
      const int which = idx >= Cn ? 1 : 0;
      const int ch = idx - which * Cn;
      const bool second = ch >= c1n;
      const int ld = second ? a.gni_ld2 : a.gni_ld1;
      const int nblk = second ? a.gni_nblk2 : a.gni_nblk1;
      const float* src = (second ? w2 + (ch - c1n) : w1 + ch) + which * ld;
      float acc = 0.f;
#pragma unroll 8
      for (int k = 0; k < nblk; ++k) acc += src[(long)k * 2 * ld];
      gt[idx] = acc;
    }
    __syncthreads();
    if (tid < a.gni_groups * 2) {
      const int g = tid >> 1, which = tid & 1;
      double acc = 0.0;
      for (int e = 0; e < cpg; ++e) acc += (double)gt[which * Cn + g * cpg + e];
      gsum[tid] = acc;
    }
    __syncthreads();
    if (tid < a.gni_groups) {
      const double n = (double)s.hw * cpg;
      const double mean = gsum[tid * 2] / n;
      double var = gsum[tid * 2 + 1] / n - mean * mean;
      if (var < 0.0) var = 0.0;
      smean[tid] = (float)mean;
      srstd[tid] = (float)(1.0 / sqrt(var + (double)a.gni_eps));
    }
    __syncthreads();
    float* const scale = gt;
    float* const shift = gt + Cn;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int ch = tid + k * 512;
      if (ch < Cn) {
        const int g = ch / cpg;
        const float sc = srstd[g] * pg[k];
        scale[ch] = sc;
        shift[ch] = pb[k] - smean[g] * sc;
      }
    }
    __syncthreads();
    // ---- the patch, normalised once per pixel, into the DMA's LDS layout
#pragma unroll
    for (int k = 0; k < GK; ++k) {
      const int u = wave + HC_NW * k;
      if (u < gni_units) {
        const int cl = (u * s.ng_magic) >> 16, grp = u - cl * s.ngrp;
        f16x8 o = raw[k];  // (zeros unless a real pixel)
        if ((gni_ok >> k) & 1u) {
          const int ch = (mlo + cl) * 32 + piece * 8;
          const f32x4 sc0 = *(const f32x4*)(scale + ch), sc1 = *(const f32x4*)(scale + ch + 4);
          const f32x4 sh0 = *(const f32x4*)(shift + ch), sh1 = *(const f32x4*)(shift + ch + 4);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float f = (float)raw[k][j] * (j < 4 ? sc0[j] : sc1[j - 4]) + (j < 4 ? sh0[j] : sh1[j - 4]);
            if (a.gni_silu) f = upk_silu(f);
            o[j] = (f16)f;
          }
        }
        *(f16x8*)((char*)smem + (size_t)(cl * s.ngrp + grp) * 1024 + lane * 16) = o;
      }
    }
  }
  STAMP(2);

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned Pl6[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) Pl6[i] = Pl[i] << 6;

  // ---- K loop
  const char* const sm = (const char*)smem;
  int have = 0;  // rounds made available so far
  auto avail = [&]() {
    // round `have` becomes readable: this wave's DMAs of it have landed (everything outstanding has), every wave is
    // past round have - 1; then the slot of round have - 1 is refilled with round have + 1
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (have >= 1 && dr.n > 0) {  // (two slots: dr is round have + 1, its slot held round have - 1)
      issue_round(dr);
      dr = next_round(dr);
    }
    ++have;
  };
  while (have < 1) avail();  // (round 0: the patch has landed — and with it the prefetch requests, whose values are dropped here)
#pragma unroll
  for (int k = 0; k < HC_PFN; ++k) asm volatile("" ::"v"(pfx[k]));
  bool done = false;
#pragma unroll 1
  while (!done) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int rnd = (int)(sI[u] >> 16);
      const int need = rnd == kDead ? total_rounds : rnd + 1;
      while (have < need) avail();
      if (rnd == kDead) {
        done = true;
        break;
      }
      // S = byte offset of (patch pixel at this tap, piece 0) inside the LDS; bit 8 of it = bit 2 of the pixel index
      const unsigned sc = ((sI[u] & 255u) << 10) + (((sI[u] >> 8) & 255u) << 6);
      f16x8 fa[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const unsigned S = Pl6[i] + sc;
        fa[i] = *(const f16x8*)(sm + (S + (lg16 ^ ((S >> 3) & 32u))));
      }
      // the item PF steps ahead: its fragment j is requested as soon as slot u's fragment j has been multiplied
      sI[u] = __builtin_amdgcn_readfirstlane(enext.y);
      if ((sI[u] >> 16) != (unsigned)kDead) last_off = __builtin_amdgcn_readfirstlane(enext.x);
      const char* sb = wb + last_off;
      enext = tq[qn];
      qn += 4;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[u][j], fa[i], acc[i][j], 0, 0, 0);
        ring[u][j] = *(const f16x8*)(sb + vo[j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  STAMP(3);

  // ---- the four K slices of each N half are summed through LDS; wave (ks, nh) finishes the fragments ks, ks + 4, ...
  // of its half
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (the ring's last refills: nothing in flight into registers)
  __builtin_amdgcn_s_barrier();                                 // every wave is out of the patch: the ring is free
  float* red = (float*)smem;                                    // [8 waves][FW fragments][64 lanes][4]
  float* cpred = (float*)((char*)smem + s.cp_off);              // [NF][2][16] GroupNorm partials of the finished fragments
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) *(f32x4*)(red + ((wave * FW + i * NJ + j) * 64 + lane) * 4) = acc[i][j];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  STAMP(5);
#pragma unroll
  for (int q = 0; q < NJ; ++q) {
    const int fw = ks + KS * q;
    const int i = fw / NJ, jj = jbase + (fw - i * NJ);
    if (jj >= NI) continue;  // (the dead fragment of an odd NI's second half; wave-uniform)
    f32x4 v = *(const f32x4*)(red + (((nh * KS + 0) * FW + fw) * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < KS; ++w) v += *(const f32x4*)(red + (((nh * KS + w) * FW + fw) * 64 + lane) * 4);
    const int m = m0 + i * 16 + lc;
    const int n = n0 + jj * 16 + lg * 4;
    if (slab) {
      if (n < a.npad && m < a.M) slab_store(slab + (unsigned)m * (unsigned)a.npad + n, v);
      continue;
    }
    const f16x4 o = Epi::Plain::put(a, rows[q], n, v + bvs[q] + rvs[q], rrs[q]);
    if (a.gn_cp) {
      f32x4 su, sq;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float x = (float)o[k];
        su[k] = Epi::row_sum16(x);
        sq[k] = Epi::row_sum16(x * x);
      }
      if (lc == 0) {
        const int f = i * NI + jj;
        *(f32x4*)(cpred + f * 32 + lg * 4) = su;
        *(f32x4*)(cpred + f * 32 + 16 + lg * 4) = sq;
      }
    }
  }
  STAMP(6);
  if (!slab && a.gn_cp) {  // (workgroup-uniform) the MI row fragments of each column fragment combined, fixed order
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int b = m0 / a.gn_hw;
    const int blk = (m0 - b * a.gn_hw) / BM;
    float* dst = a.gn_cp + (long)((b * a.gn_nblk + blk) * 2) * a.npad;
    if (lane < 32) {
      const int which = lane >> 4, col = lane & 15;
      for (int j = wave; j < NI; j += NW) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) t += cpred[(i * NI + j) * 32 + which * 16 + col];
        const int n = n0 + j * 16 + col;
        if (n < a.npad) dst[which * a.npad + n] = t;
      }
    }
  }
#ifdef UPK_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(4);
#endif
#undef STAMP
}

struct HcCfg {
  int mi, ni, pf;
  const char* name;
  void (*fn)(const HcArgs, const IgemmArgs);
  void (*fn_gni)(const HcArgs, const IgemmArgs);
};
#define HCC(MI, NI, PF) {MI, NI, PF, "hc" #MI "x" #NI "p" #PF, halo_conv_kernel<MI, NI, PF, false>, halo_conv_kernel<MI, NI, PF, true>}
const HcCfg kHc[] = {
    HCC(4, 7, 4),  // 64 x 112 (the 7 * 32 channel family): waves hold 64 x 64 | 64 x 48, 16 KiB of weights in flight per wave
    HCC(4, 4, 8),  // 64 x 64: waves hold 64 x 32, 16 KiB in flight
    // (128 x 64 tiles — half the weight bytes per output pixel — measured: 16.2-16.9 us against 17.2 standalone, no shape
    // taken in situ: DESIGN.md 11d; the epilogue below finishes NJ fragments per wave, i.e. MI = 4 only)
};
constexpr int kNumHc = sizeof(kHc) / sizeof(kHc[0]);
unsigned long long hc_attr_set[kNumHc][2];  // (per device: upk_lds_attr_once)

int lg2i(int v) {
  int sft = 0;
  while ((1 << sft) < v) ++sft;
  return (1 << sft) == v ? sft : -1;
}

}  // namespace

int hc_num_configs() { return kNumHc; }
const char* hc_config_name(int c) { return (c >= 0 && c < kNumHc) ? kHc[c].name : "?"; }



// Whether configuration c with `splitk` K splits takes the launch described by `a` (already filled by conv_impl up to
// the tile counts); fills the plan.
bool hc_plan(const upk_ctx* ctx, const IgemmArgs& a, int c, int splitk, HcPlan* pl) {
  (void)ctx;
  if (c < 0 || c >= kNumHc || !pl) return false;
  if (a.ks != 3 || a.stride != 1 || a.ups || a.ph_on || a.pad_lo != 1 || a.ln_u) return false;
  if (a.Ho != a.HS || a.Wo != a.WS) return false;
  const int H = a.HS, W = a.WS, hw = H * W;
  const int shw = lg2i(hw), sw = lg2i(W);
  const int BM = kHc[c].mi * 16;
  if (shw < 0 || sw < 0 || W < 4 || W > 64 || a.M % BM) return false;  // (BM | M: tiles are full)
  if (hw >= BM ? (hw % BM != 0) : (BM % hw != 0)) return false;
  if ((a.c1 | a.c2 | a.c3 | a.c4) & 31) return false;
  if (splitk < 1) splitk = 1;
  const bool slabs = splitk > 1;
  if (!slabs && !Epi::plain(a)) return false;
  if (a.lnr_out || a.lnr_in) return false;
  const int rows_part = hw >= BM ? BM / W : H;
  const int nparts = hw >= BM ? 1 : BM / hw;
  const int pw = W + 2;
  const int part_pix = (rows_part + 2) * pw;
  const int npix = nparts * part_pix;
  const int ngrp = (npix + 15) / 16;
  if (ngrp > 16) return false;  // (two DMA groups per wave)
  const int cpt = (a.c1 + a.c2) / 32, capp = (a.c3 + a.c4) / 32;
  const int mps = (cpt + splitk - 1) / splitk, aps = (capp + splitk - 1) / splitk;
  if (splitk > 1 && (mps < 1 || (splitk - 1) * mps >= cpt)) return false;  // (every split has 3x3 chunks)
  // rounds: the whole K range of a workgroup in one slot when it fits, else two slots of cr chunks
  const int NF = kHc[c].mi * kHc[c].ni;
  const int red_bytes = HC_NW * kHc[c].mi * ((kHc[c].ni + 1) / 2) * 1024;  // [8 waves][MI x NJ fragments][1 KiB]
  const int budget = 148 * 1024;  // (160 KiB less the GroupNorm scratch and the K-item table)
  const int kchunks = mps + aps;
  int cr, nslot;
  if (kchunks * ngrp * 1024 <= budget) {
    cr = kchunks > 0 ? kchunks : 1;
    nslot = 1;
  } else {
    cr = (budget / 2) / (ngrp * 1024);
    nslot = 2;
    if (cr < 1) return false;
  }
  const int ring_bytes = nslot * cr * ngrp * 1024;
  pl->cp_off = ring_bytes > red_bytes ? ring_bytes : red_bytes;
  pl->tab_off = pl->cp_off + NF * 128;
  pl->lds_bytes = pl->tab_off + (mps * 9 + aps + 4 * (kHc[c].pf + 2)) * 8;
  pl->gni = 0;
  pl->gni_off = 0;
  if (a.gni_st1) {
    // input GroupNorm in the patch fill: the whole K range of a workgroup in one slot (every raw load is issued up
    // front), one sample per tile, <= HC_GNI_K (chunk, group) units per wave, statistics as <= 32 row blocks per source
    const int Cn = a.c1 + a.c2;
    if (nslot != 1 || hw < BM || mps * ngrp > HC_NW * HC_GNI_K || Cn > 1024) return false;
    if (a.gni_groups <= 0 || a.gni_groups > UPK_GN_GROUPS_MAX || Cn % a.gni_groups || !a.gni_gamma || !a.gni_beta) return false;
    if (a.gni_nblk1 <= 0 || a.gni_nblk1 > UPK_GN_MAX_CHUNKS || a.gni_ld1 < a.c1) return false;
    if (a.c2 && (!a.gni_st2 || a.gni_nblk2 <= 0 || a.gni_nblk2 > UPK_GN_MAX_CHUNKS || a.gni_ld2 < a.c2)) return false;
    pl->gni = 1;
    pl->gni_off = (pl->lds_bytes + 15) & ~15;
    pl->lds_bytes = pl->gni_off + 2 * Cn * 4 + 2 * UPK_GN_GROUPS_MAX * 8 + 2 * UPK_GN_GROUPS_MAX * 4;
  }
  if (pl->lds_bytes > 160 * 1024) return false;
  pl->bn = kHc[c].ni * 16;
  pl->bm = BM;
  pl->splitk = splitk;
  pl->pw = pw, pl->part_pix = part_pix, pl->npix = npix, pl->ngrp = ngrp, pl->cr = cr, pl->nslot = nslot;
  pl->cpt = cpt, pl->mps = mps, pl->aps = aps, pl->sh_hw = shw, pl->sh_w = sw;
  pl->pp_magic = (65536 + part_pix - 1) / part_pix, pl->pw_magic = (65536 + pw - 1) / pw;
  return true;
}

int hc_launch(upk_ctx* ctx, const IgemmArgs& a, int c, const HcPlan& pl, dim3 grid, hipStream_t stream) {
  HcArgs s;
  memset(&s, 0, sizeof(s));
  s.x1 = a.x1, s.x2 = a.x2, s.x3 = a.x3, s.x4 = a.x4, s.w = a.w, s.zero = a.zero, s.partial = a.partial;
  s.ch1 = a.c1 / 32, s.ch2 = a.c2 / 32, s.ch3 = a.c3 / 32, s.ch4 = a.c4 / 32;
  s.ld1 = a.ld1, s.ld2 = a.ld2, s.ld3 = a.ld3, s.ld4 = a.ld4;
  s.npad = a.npad, s.M = a.M, s.H = a.HS, s.W = a.WS;
  s.pw = pl.pw, s.part_pix = pl.part_pix, s.hw = a.HS * a.WS, s.npix = pl.npix, s.ngrp = pl.ngrp;
  s.cr = pl.cr, s.nslot = pl.nslot, s.cpt = pl.cpt, s.mps = pl.mps, s.aps = pl.aps, s.cp_off = pl.cp_off;
  s.sh_hw = pl.sh_hw, s.sh_w = pl.sh_w, s.tab_off = pl.tab_off, s.pp_magic = pl.pp_magic, s.pw_magic = pl.pw_magic;
  s.gni_off = pl.gni_off;
  s.ng_magic = (65536 + pl.ngrp - 1) / pl.ngrp;
  const auto fn = pl.gni ? kHc[c].fn_gni : kHc[c].fn;
  if (int rc = upk_lds_attr_once(ctx, (const void*)fn, &hc_attr_set[c][pl.gni])) return rc;
  hipLaunchKernelGGL(fn, grid, dim3(512), (size_t)pl.lds_bytes, stream, s, a);
  return upk_check_launch(ctx, "halo_conv");
}

}  // namespace upkd

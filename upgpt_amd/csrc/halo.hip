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
// split-K over channel ranges with the usual fp32 slabs (the reduce passes of igemm.hip follow); plain epilogue (bias,
// timestep row vector, residual -> fp16 NHWC, GroupNorm channel partials as a by-product) or slabs.
#include "igemm_common.h"

namespace upkd {
namespace {

constexpr int HC_NW = 8;
constexpr int HC_MI = 4;

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
  int sh_hw, sh_w;  // log2 of H * W / W (powers of two by construction)
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

template <int NI, int PF>
__global__ __launch_bounds__(512) void halo_conv_kernel(const HcArgs s, const IgemmArgs a) {
  constexpr int MI = HC_MI, NW = HC_NW, NF = MI * NI;
  constexpr int G = NF <= 16 ? NF : NF / 2;  // fragments per reduction group (G * 8 KiB of LDS)
  static_assert(NF % G == 0, "reduction groups");
  extern __shared__ __attribute__((aligned(16))) f16 smem[];

  HC_PIN(s.x1); HC_PIN(s.x2); HC_PIN(s.x3); HC_PIN(s.x4); HC_PIN(s.w); HC_PIN(s.zero); HC_PIN(s.partial);
  HC_PIN(s.ch1); HC_PIN(s.ch2); HC_PIN(s.ch3); HC_PIN(s.ch4); HC_PIN(s.ld1); HC_PIN(s.ld2); HC_PIN(s.ld3); HC_PIN(s.ld4);
  HC_PIN(s.npad); HC_PIN(s.M); HC_PIN(s.H); HC_PIN(s.W); HC_PIN(s.pw); HC_PIN(s.part_pix); HC_PIN(s.hw); HC_PIN(s.npix);
  HC_PIN(s.ngrp); HC_PIN(s.cr); HC_PIN(s.nslot); HC_PIN(s.cpt); HC_PIN(s.mps); HC_PIN(s.aps); HC_PIN(s.cp_off);
  HC_PIN(s.sh_hw); HC_PIN(s.sh_w);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  const int m0 = tm * 64;
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
  // (a) as a DMA lane: patch pixel P = group * 16 + lane / 4 of the groups {wave, wave + 8}, piece (lane & 3) swizzled
  const int b0 = m0 >> s.sh_hw;                          // first image of the tile
  const int y0 = (m0 - (b0 << s.sh_hw)) >> s.sh_w;       // first output row inside it (0 when the tile holds whole images)
  const int piece = (lane & 3) ^ (((lane >> 4) & 1) << 1);
  int pix[2];   // linear input pixel (b * H + iy) * W + ix of the lane's patch pixel, or -1 (padding / past the patch)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int P = (wave + 8 * q) * 16 + (lane >> 2);
    const int part = P / s.part_pix, rem = P - part * s.part_pix;
    const int py = rem / s.pw, px = rem - py * s.pw;
    const int iy = y0 + py - 1, ix = px - 1;
    const int b = b0 + part;
    const bool ok = P < s.npix && iy >= 0 && iy < s.H && ix >= 0 && ix < s.W && (b << s.sh_hw) < s.M;
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
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int grp = wave + 8 * q;
      if (grp >= s.ngrp) break;  // (wave-uniform)
      const f16* src = pix[q] >= 0 ? sp + (long)pix[q] * sld + c0 + piece * 8 : zsrc;
      const int step = pix[q] >= 0 ? 32 : 0;
      for (int cl = 0; cl < r.n; ++cl)
        hc_dma16(src + cl * step, dst0 + (unsigned)((cl * s.ngrp + grp) * 1024));
    }
  };
  // (b) as an MFMA lane: patch pixel of tile pixel t = i * 16 + lc at tap (0, 0)
  unsigned Pl[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int t = i * 16 + lc;
    const int part = t >> s.sh_hw;  // (0 unless the tile holds whole images: H * W < 64)
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
    issue_round(dr);
    dr = next_round(dr);
  }
  STAMP(1);

  // ---- weight ring.  Cursor = the item PF steps ahead of the one being multiplied; its A-side coordinates travel with
  // the ring slot (sA: LDS byte offset of the chunk, sT: tap pixel offset, sR: round index; sR == kDead: no item)
  const int kDead = 1 << 20;
  HcRound cr_ = r_first;  // round of the cursor
  int c_next = wave;      // local item index of this wave's next item inside cr_
  const char* const wb = (const char*)s.w + (size_t)n0 * 64u;
  const unsigned kstr = (unsigned)s.npad * 64u;  // bytes per K chunk of the packed weight
  const unsigned voff = (unsigned)lane * 16u;
  unsigned jo[NI];  // byte offset of fragment j inside the tile's run (a fragment past n_pad re-reads fragment 0: its outputs are never stored)
#pragma unroll
  for (int j = 0; j < NI; ++j) jo[j] = n0 + j * 16 < s.npad ? (unsigned)j * 1024u : 0u;
  f16x8 ring[PF][NI];
  unsigned sA[PF];
  int sT[PF], sR[PF];
  auto gen = [&](f16x8 (&dstring)[NI], unsigned& oA, int& oT, int& oR) {
    // move to the round that holds local item c_next
    int cnt = cr_.n * (cr_.g >= cpt ? 1 : 9);
    while (cr_.n > 0 && c_next >= cnt) {
      c_next -= cnt;
      cr_ = next_round(cr_);
      cnt = cr_.n * (cr_.g >= cpt ? 1 : 9);
    }
    const bool live = cr_.n > 0;
    const bool app = cr_.g >= cpt;
    const int cl = live ? (app ? c_next : (c_next * 7282) >> 16) : 0;
    const int tap = live ? (app ? 4 : c_next - 9 * cl) : 0;
    const int ky = (tap * 43) >> 7, kx = tap - 3 * ky;
    const int kc = live ? (app ? 9 * cpt + (cr_.g - cpt) + cl : tap * cpt + cr_.g + cl) : 0;
    const char* sb = wb + (size_t)((unsigned)kc * kstr);
#pragma unroll
    for (int j = 0; j < NI; ++j) dstring[j] = *(const f16x8*)(sb + jo[j] + voff);
    oA = (unsigned)((cr_.base + cl) * s.ngrp) * 1024u;
    oT = ky * s.pw + kx;
    oR = live ? cr_.idx : kDead;
    c_next += NW;
  };
#pragma unroll
  for (int u = 0; u < PF; ++u) gen(ring[u], sA[u], sT[u], sR[u]);
  STAMP(2);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

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
  bool done = false;
#pragma unroll 1
  while (!done) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int need = sR[u] == kDead ? total_rounds : sR[u] + 1;
      while (have < need) avail();
      if (sR[u] == kDead) {
        done = true;
        break;
      }
      f16x8 fa[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const unsigned t = Pl[i] + (unsigned)sT[u];
        const unsigned ad = sA[u] + (t << 6) + (lg16 ^ ((t & 4u) << 3));
        fa[i] = *(const f16x8*)(sm + ad);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[u][j], fa[i], acc[i][j], 0, 0, 0);
      gen(ring[u], sA[u], sT[u], sR[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  STAMP(3);

  // ---- the eight K slices are summed through LDS, G fragments at a time; wave w finishes fragments f0 + w, f0 + w + 8
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (dead refills of the ring: nothing in flight into registers)
  __builtin_amdgcn_s_barrier();                                 // every wave is out of the patch: the ring is free
  float* red = (float*)smem;                                    // [8 waves][G fragments][64 lanes][4]
  float* cpred = (float*)((char*)smem + s.cp_off);              // [NF][2][16] GroupNorm partials of the finished fragments
  constexpr int FQ = (G + NW - 1) / NW;                         // fragments a wave finishes per group
  const Epi::Plain P(a);
  float* slab = s.partial ? s.partial + (long)zs * a.M * a.npad : nullptr;
#pragma unroll
  for (int f0 = 0; f0 < NF; f0 += G) {
#pragma unroll
    for (int q = 0; q < G; ++q) {
      const int f = f0 + q;
      *(f32x4*)(red + ((wave * G + q) * 64 + lane) * 4) = acc[f / NI][f % NI];
    }
    // operands of this wave's fragments (requested before the barrier: one round trip under the LDS exchange)
    Epi::Plain::Row rows[FQ];
    f32x4 bvs[FQ], rvs[FQ];
    f16x4 rrs[FQ];
    if (!slab) {
#pragma unroll
      for (int q = 0; q < FQ; ++q) {
        const int fl = wave + NW * q;
        const int f = f0 + fl;
        const int i = f / NI, j = f - i * NI;
        const int n = n0 + j * 16 + lg * 4;
        rows[q] = P.row(a, fl < G ? m0 + i * 16 + lc : a.M, a.M);
        bvs[q] = P.bias4(a, n);
        rvs[q] = P.rv4(a, rows[q], n);
        rrs[q] = P.res4(a, rows[q], n);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < FQ; ++q) {
      const int fl = wave + NW * q;
      if (fl >= G) continue;  // (wave-uniform)
      const int f = f0 + fl;
      const int i = f / NI, j = f - i * NI;
      f32x4 v = *(const f32x4*)(red + ((0 * G + fl) * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < NW; ++w) v += *(const f32x4*)(red + ((w * G + fl) * 64 + lane) * 4);
      const int m = m0 + i * 16 + lc;
      const int n = n0 + j * 16 + lg * 4;
      if (slab) {
        if (n < a.npad && m < a.M) *(f32x4*)(slab + (unsigned)m * (unsigned)a.npad + n) = v;
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
          *(f32x4*)(cpred + f * 32 + lg * 4) = su;
          *(f32x4*)(cpred + f * 32 + 16 + lg * 4) = sq;
        }
      }
    }
    if (f0 + G < NF) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // (the reduction buffer is rewritten by the next group)
    }
  }
  if (!slab && a.gn_cp) {  // (workgroup-uniform) the MI row fragments of each column fragment combined, fixed order
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int b = m0 / a.gn_hw;
    const int blk = (m0 - b * a.gn_hw) / 64;
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
  int ni, pf;
  const char* name;
  void (*fn)(const HcArgs, const IgemmArgs);
};
#define HCC(NI, PF) {NI, PF, "hc" #NI "p" #PF, halo_conv_kernel<NI, PF>}
const HcCfg kHc[] = {
    HCC(7, 2),  // 64 x 112 (the 7 * 32 channel family), 14 KiB of weights in flight per wave
    HCC(4, 4),  // 64 x 64, 16 KiB in flight
    HCC(4, 2),  // 64 x 64,  8 KiB in flight (short K ranges per split)
    HCC(8, 2),  // 64 x 128 (power-of-two channel counts: the upscale UNet)
};
constexpr int kNumHc = sizeof(kHc) / sizeof(kHc[0]);
bool hc_attr_done[kNumHc];

int lg2i(int v) {
  int sft = 0;
  while ((1 << sft) < v) ++sft;
  return (1 << sft) == v ? sft : -1;
}

}  // namespace

int hc_num_configs() { return kNumHc; }
const char* hc_config_name(int c) { return (c >= 0 && c < kNumHc) ? kHc[c].name : "?"; }
int hc_config_bn(int c) { return (c >= 0 && c < kNumHc) ? kHc[c].ni * 16 : 0; }

// Whether configuration c with `splitk` K splits takes the launch described by `a` (already filled by conv_impl up to
// the tile counts); fills the plan.
bool hc_plan(const upk_ctx* ctx, const IgemmArgs& a, int c, int splitk, HcPlan* pl) {
  (void)ctx;
  if (c < 0 || c >= kNumHc || !pl) return false;
  if (a.ks != 3 || a.stride != 1 || a.ups || a.ph_on || a.pad_lo != 1 || a.ln_u) return false;
  if (a.Ho != a.HS || a.Wo != a.WS) return false;
  const int H = a.HS, W = a.WS, hw = H * W;
  const int shw = lg2i(hw), sw = lg2i(W);
  if (shw < 0 || sw < 0 || W < 4 || W > 64 || a.M % 64) return false;  // (64 | M: tiles are full)
  if (hw >= 64 ? (hw % 64 != 0) : (64 % hw != 0)) return false;
  if ((a.c1 | a.c2 | a.c3 | a.c4) & 31) return false;
  if (splitk < 1) splitk = 1;
  const bool slabs = splitk > 1;
  if (!slabs && !Epi::plain(a)) return false;
  if (a.lnr_out || a.lnr_in) return false;
  const int rows_part = hw >= 64 ? 64 / W : H;
  const int nparts = hw >= 64 ? 1 : 64 / hw;
  const int pw = W + 2;
  const int part_pix = (rows_part + 2) * pw;
  const int npix = nparts * part_pix;
  const int ngrp = (npix + 15) / 16;
  if (ngrp > 16) return false;  // (two DMA groups per wave)
  const int cpt = (a.c1 + a.c2) / 32, capp = (a.c3 + a.c4) / 32;
  const int mps = (cpt + splitk - 1) / splitk, aps = (capp + splitk - 1) / splitk;
  if (splitk > 1 && (mps < 1 || (splitk - 1) * mps >= cpt)) return false;  // (every split has 3x3 chunks)
  // rounds: the whole K range of a workgroup in one slot when it fits, else two slots of cr chunks
  const int NF = HC_MI * kHc[c].ni;
  const int Gf = NF <= 16 ? NF : NF / 2;
  const int red_bytes = HC_NW * Gf * 1024;
  const int budget = 152 * 1024;
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
  pl->lds_bytes = pl->cp_off + NF * 128;
  if (pl->lds_bytes > 160 * 1024) return false;
  pl->bn = kHc[c].ni * 16;
  pl->splitk = splitk;
  pl->pw = pw, pl->part_pix = part_pix, pl->npix = npix, pl->ngrp = ngrp, pl->cr = cr, pl->nslot = nslot;
  pl->cpt = cpt, pl->mps = mps, pl->aps = aps, pl->sh_hw = shw, pl->sh_w = sw;
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
  s.sh_hw = pl.sh_hw, s.sh_w = pl.sh_w;
  if (!hc_attr_done[c]) {  // (one call per configuration and process: a no-op on current ROCm, kept for runtimes that honour it)
    UPK_HIP(ctx, hipFuncSetAttribute((const void*)kHc[c].fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hc_attr_done[c] = true;
  }
  hipLaunchKernelGGL(kHc[c].fn, grid, dim3(512), (size_t)pl.lds_bytes, stream, s, a);
  return upk_check_launch(ctx, "halo_conv");
}

}  // namespace upkd

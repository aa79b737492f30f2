// A-stationary "patch" convolution / Linear for gfx950 (CDNA4), fp16 in, fp32 accumulate.
//
// The implicit-GEMM kernels of igemm.hip stream BOTH operands through the LDS-DMA ring, so a 3x3 conv
// re-fetches every input pixel nine times through the per-CU L2 -> LDS fill path (~36 B/clk/CU), which is
// what bounds their K loops, and the GroupNorm + SiLU in front of every ResBlock conv (openaimodel.py:255-275)
// has to be a launch of its own because nothing ever sees the A operand in registers.
//
// Here an M tile is a set of whole output image rows of ONE sample.  Its input patch (tile rows + the 3x3 halo,
// a slab of <= p_cs channels at a time) is staged ONCE per slab in LDS *through registers* by all eight waves —
// which is where the producer GroupNorm's per-(sample, channel) affine and the SiLU are applied, zero padding
// after them, exactly as F.conv2d sees the normalised tensor — and the four MFMA waves then walk the ksize^2
// taps of every 32-channel chunk straight out of that patch (tap = a scalar byte offset on the fragment address).
// Only the weights stream: global -> LDS by direct-to-LDS DMA into a ring of NBUF stage slots (the loader half of
// igemm_ws_kernel), so fill traffic per workgroup is  patch + BN * K * 2 B  instead of  (BM + BN) * K * 2 B.
//
//   LDS:  [ weight ring NBUF x KS x BN x 64 B | 1 KiB dump | 1 KiB statistics | patch p_np x p_ps B ]
//   patch pixel stride p_ps = 2 * p_cs + 32 B (32 mod 64): ds_read_b128 of 16 consecutive pixels x 2 k-slices is
//   bank-conflict free for every tap shift (checked exhaustively); patch rows are p_pw pixels wide (Wo + 2, or
//   Wo + 8 when a 16-row fragment spans several image rows).
//
// Wave roles: waves 0..3 MFMA consumers = WM row groups x KW K-slices (KW > 1: the slices of a stage are dealt
// round-robin and summed through LDS at the end, fixed order), waves 4..7 weight loaders; everybody stages.
// GroupNorm statistics of the INPUT come from partial sums the producer launch left behind (include/upk.h
// gni_mode), folded per workgroup in a fixed order (fp64) -> bitwise reproducible.
// Epilogues are the ones of igemm.hip (igemm_common.h), incl. the GroupNorm partials of the OUTPUT.
#include "igemm_common.h"

namespace {
using namespace upkd;

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

template <int N>
__device__ __forceinline__ void pc_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// LDS writes of this wave are visible to the workgroup after the barrier
__device__ __forceinline__ void pc_bar() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Walks the chunk sequence of the launch: slab by slab, inside a slab tap-major, padded with dead chunks to a
// whole number of KS-chunk stages.  All members are wave-uniform.
template <int KS>
struct PcCursor {
  int slab, q, tap, c;          // position: slab, chunk index inside the slab (incl. dead ones), tap, chunk of the tap
  int cpsl, nchunk, nq, kcbase;  // this slab: 32-chunks per tap, live chunks, padded chunks, first weight chunk
  bool app;
  __device__ __forceinline__ void enter(const IgemmArgs& a, int s, int nslab_main) {
    slab = s;
    app = s >= nslab_main;
    const int si = app ? s - nslab_main : s;
    const int c0 = si * a.p_cs;
    const int tot = app ? a.c3 + a.c4 : a.c1 + a.c2;
    const int csl = min(a.p_cs, tot - c0);
    cpsl = csl >> 5;
    nchunk = (app ? 1 : a.ks * a.ks) * cpsl;
    nq = (nchunk + KS - 1) / KS * KS;
    kcbase = (app ? a.nchunks_main : 0) + (c0 >> 5);
    q = tap = c = 0;
  }
  __device__ __forceinline__ bool live() const { return q < nchunk; }
  __device__ __forceinline__ int kc(const IgemmArgs& a) const { return kcbase + tap * a.cpt + c; }
  __device__ __forceinline__ void next() {
    ++q;
    if (++c == cpsl) {
      c = 0;
      ++tap;
    }
  }
};

template <int MI, int NI, int WM, int KW, int KS, int NBUF>
__global__ __launch_bounds__(512) void pconv_kernel(const IgemmArgs a) {
  static_assert(WM * KW == 4, "4 MFMA waves");
  static_assert(KS % KW == 0, "every K slice gets the same number of chunks per stage");
  constexpr int BN = NI * 16;
  constexpr int BGW = (NI + 3) / 4;  // 16-row weight groups per loader wave
  constexpr int P = KS * BGW;        // DMAs per loader wave per stage
  constexpr int D = NBUF - 1;
  static_assert(D * P <= 63, "vmcnt range");
  constexpr int STAGE = KS * BN * 32;  // halfs per ring slot
  constexpr int RING_BYTES = NBUF * STAGE * 2;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f16* const ring = (f16*)smem;
  f16* const dump = ring + NBUF * STAGE;
  float* const gstat = (float*)(smem + RING_BYTES + 1024);         // mean[32], rstd[32]
  double* const gsum = (double*)(smem + RING_BYTES + 1024 + 256);  // [64]
  unsigned char* const patch = smem + RING_BYTES + 2048;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x;
  const int tn = tile / a.tiles_m;
  const int tm = tile - tn * a.tiles_m;
  const int m0 = tm * a.tile_rows;
  const int n0 = tn * BN;
  const int hw = a.Ho * a.Wo;
  const int bsmp = m0 / hw;
  const int y0 = (m0 - bsmp * hw) / a.Wo;
  const bool k3 = a.ks == 3;
  const int PW = a.p_pw, PS = a.p_ps;
  const int ctot = a.c1 + a.c2, capp = a.c3 + a.c4;
  const int nslab_main = (ctot + a.p_cs - 1) / a.p_cs;
  const int nslab = nslab_main + (capp + a.p_cs - 1) / a.p_cs;
  const bool is_loader = wave >= 4;

  // total number of stages (uniform)
  int T = 0;
  {
    PcCursor<KS> t;
    for (int s = 0; s < nslab; ++s) {
      t.enter(a, s, nslab_main);
      T += t.nq / KS;
    }
  }

  // ------------------------------------------------ loader state: weight cursor + the first D stages
  const int lw = wave - 4;
  const int r16 = lane >> 2;
  const int chd = (lane & 3) ^ ((-(lane >> 4)) & 3);  // source chunk of this lane (XOR swizzle on the source)
  const f16* zsrc = a.zero + (lane & 3) * 8;
  const f16* wl[BGW];
  bool b_ok[BGW];
  PcCursor<KS> bc;
  const long wstep = (long)a.npad * 32;
  int issued = 0;
  auto issue_stage = [&](int slot) {
    f16* base = ring + slot * STAGE;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bool live = bc.slab < nslab && bc.live();
      const long koff = live ? (long)bc.kc(a) * wstep : 0;
#pragma unroll
      for (int i = 0; i < BGW; ++i) {
        const int rg = lw + 4 * i;  // wave-uniform
        const f16* src = (live && b_ok[i]) ? wl[i] + koff : zsrc;
        f16* dst = (rg < NI) ? base + (s * BN + rg * 16) * 32 : dump;
        __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)dst, 16, 0, 0);
      }
      if (bc.slab < nslab) {
        bc.next();
        if (bc.q == bc.nq) {
          if (bc.slab + 1 < nslab) bc.enter(a, bc.slab + 1, nslab_main);
          else bc.slab = nslab;
        }
      }
    }
  };
  auto wait_oldest = [&](int r) {  // at most r - 1 whole stages stay in flight
    static_assert(D <= 3, "wait ladder");
    if (D >= 3 && r >= 3) pc_wait_vmcnt<(D >= 3 ? 2 : 0) * P>();
    else if (D >= 2 && r == 2) pc_wait_vmcnt<P>();
    else pc_wait_vmcnt<0>();
  };
  if (is_loader) {
#pragma unroll
    for (int i = 0; i < BGW; ++i) {
      const int rg = lw + 4 * i;
      const int row = rg * 16 + r16;
      b_ok[i] = (rg < NI) && (n0 + row < a.npad);
      wl[i] = a.w + (long)(n0 + row) * 32 + chd * 8;
    }
    bc.enter(a, 0, nslab_main);
    for (; issued < D && issued < T; ++issued) issue_stage(issued % NBUF);
  }

  // ------------------------------------------------ GroupNorm statistics of the input (all waves)
  if (a.gni_mode) {
    const int groups = a.gni_groups, cpg = a.gni_cpg;
    if (a.gni_mode == 2) {
      float* chs = (float*)patch;  // [2][ctot] channel sums; the patch is staged after the last read
      const float* w1 = a.gni_s1 + (long)bsmp * a.gni_nblk1 * 2 * a.gni_ld1;
      const float* w2 = a.gni_s2 ? a.gni_s2 + (long)bsmp * a.gni_nblk2 * 2 * a.gni_ld2 : nullptr;
      for (int idx = tid; idx < 2 * ctot; idx += 512) {
        const int which = idx >= ctot ? 1 : 0;
        const int ch = idx - which * ctot;
        const bool second = ch >= a.c1;
        const int ld = second ? a.gni_ld2 : a.gni_ld1;
        const int nblk = second ? a.gni_nblk2 : a.gni_nblk1;
        const float* src = (second ? w2 + (ch - a.c1) : w1 + ch) + which * ld;
        float acc = 0.f;
#pragma unroll 4
        for (int k = 0; k < nblk; ++k) acc += src[(long)k * 2 * ld];
        chs[idx] = acc;
      }
      pc_bar();
      if (tid < groups * 2) {
        const int g = tid >> 1, which = tid & 1;
        double acc = 0.0;
        for (int e = 0; e < cpg; ++e) acc += (double)chs[which * ctot + g * cpg + e];
        gsum[tid] = acc;
      }
      pc_bar();
    } else {
      if (tid < groups * 2) {  // tid = group * 2 + {sum, sumsq}
        const float* w = a.gni_s1 + (long)bsmp * a.gni_nblk1 * groups * 2 + tid;
        double acc = 0.0;
        for (int k = 0; k < a.gni_nblk1; ++k) acc += (double)w[(long)k * groups * 2];
        gsum[tid] = acc;
      }
      pc_bar();
    }
    if (tid < groups) {
      const double n = (double)hw * cpg;
      const double mean = gsum[tid * 2] / n;
      double var = gsum[tid * 2 + 1] / n - mean * mean;
      if (var < 0.0) var = 0.0;
      gstat[tid] = (float)mean;
      gstat[32 + tid] = (float)(1.0 / sqrt(var + (double)a.gni_eps));
    }
    pc_bar();
  }

  // ------------------------------------------------ staging pass of one slab (all waves)
  auto stage_A = [&](bool app, int c0, int csl) {
    const int tl = a.p_tpp_log2;
    const int v = tid & ((1 << tl) - 1);
    const int slot = tid >> tl;
    const int nslots = 512 >> tl;
    const bool vact = v < (csl >> 3);
    const int c = c0 + v * 8;
    const int cf = app ? a.c3 : a.c1;
    const bool second = c >= cf;
    const f16* sbase = app ? (second ? a.x4 + (c - cf) : a.x3 + c) : (second ? a.x2 + (c - cf) : a.x1 + c);
    const long sld = app ? (second ? a.ld4 : a.ld3) : (second ? a.ld2 : a.ld1);
    const bool xf = !app && a.gni_mode != 0;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = 1.f;
      sh[j] = 0.f;
    }
    if (xf && vact) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ch = c + j;
        const int g = ch / a.gni_cpg;
        const float s = gstat[32 + g] * a.gni_gamma[ch];
        sc[j] = s;
        sh[j] = a.gni_beta[ch] - gstat[g] * s;
      }
    }
    const bool silu = xf && a.gni_silu;
    const int np = a.p_np;
    // incremental (patch row, patch column) of this thread's pixels: pix = slot + it * nslots
    int ry, cx;
    int sdiv, smod;
    if (k3) {
      ry = slot / PW;
      cx = slot - ry * PW;
      sdiv = nslots / PW;
      smod = nslots - sdiv * PW;
    } else {
      ry = slot;
      cx = 0;
      sdiv = nslots;
      smod = 0;
    }
    const int R = a.tile_rows / a.Wo;  // image rows of the tile (3x3)
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int U = 4;
    const int nit = (np + nslots - 1) / nslots;
#pragma unroll 1
    for (int it = 0; it < nit; it += U) {
      f16x8 xv[U];
      int doff[U];
      int st[U];  // 0: nothing to store, 1: zeros (padding), 2: value
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pix = slot + (it + u) * nslots;
        st[u] = 0;
        xv[u] = zero8;
        doff[u] = pix * PS + v * 16;
        if (vact && pix < np) {
          long gp;
          bool inimg;
          bool need = true;
          if (k3) {
            const int iy = y0 - 1 + ry, ix = cx - 1;
            inimg = iy >= 0 && iy < a.HS && ix >= 0 && ix < a.WS;
            need = cx < a.WS + 2;
            if (app) need = need && ry >= 1 && ry <= R && cx >= 1 && cx <= a.WS;  // only the centre tap reads it
            gp = ((long)bsmp * a.HS + iy) * a.WS + ix;
          } else {
            const int m = m0 + pix;
            inimg = pix < a.tile_rows && m < a.M;
            gp = m;
          }
          if (need) {
            st[u] = inimg ? 2 : 1;
            if (inimg) xv[u] = *(const f16x8*)(sbase + gp * sld);
          }
        }
        // next pixel of this thread
        ry += sdiv;
        cx += smod;
        if (k3 && cx >= PW) {
          cx -= PW;
          ++ry;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (st[u] == 0) continue;
        f16x8 y = xv[u];
        if (xf && st[u] == 2) {
          float f[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = (float)xv[u][j] * sc[j] + sh[j];
          if (silu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = upk_silu(f[j]);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) y[j] = (f16)f[j];
        }
        *(f16x8*)(patch + doff[u]) = y;
      }
    }
  };

  // ------------------------------------------------ consumer state
  const int wm = wave % WM, kw = (wave & 3) / WM;
  const int lg = lane >> 4, lc = lane & 15;
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int aoff0[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    int mloc = (wm * MI + i) * 16 + lc;
    if (mloc >= a.tile_rows) mloc = 0;  // rows past the tile's valid range read pixel 0 (results are masked)
    int pix0 = mloc;
    if (k3) {
      const int r = mloc / a.Wo;
      pix0 = r * PW + (mloc - r * a.Wo);
    }
    aoff0[i] = pix0 * PS + lg * 16;
  }
  const int frag_off = lc * 32 + lds_swz(lc, lg) * 8;
  const int ctr = k3 ? (PW + 1) * PS : 0;  // byte offset of the centre tap

  // ------------------------------------------------ slabs
  int tglob = 0;
  PcCursor<KS> cc;
  for (int slab = 0; slab < nslab; ++slab) {
    cc.enter(a, slab, nslab_main);
    {
      const int si = cc.app ? slab - nslab_main : slab;
      const int c0 = si * a.p_cs;
      stage_A(cc.app, c0, cc.cpsl * 32);
    }
    if (is_loader && slab == 0) wait_oldest(issued);  // stage 0 landed
    pc_bar();
    const int nst = cc.nq / KS;
    if (is_loader) {
      for (int t = 0; t < nst; ++t) {
        if (issued < T) {
          issue_stage(issued % NBUF);
          ++issued;
        }
        const int outstanding = issued - (tglob + t + 1);
        if (outstanding > 0) wait_oldest(outstanding);
        __builtin_amdgcn_s_barrier();
      }
    } else {
      for (int t = 0; t < nst; ++t) {
        const f16* slot = ring + ((tglob + t) % NBUF) * STAGE;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          if (cc.live() && (KW == 1 || (s % KW) == kw)) {
            int tapb = ctr;
            if (k3 && !cc.app) {
              const int ky = (cc.tap * 11) >> 5;
              const int kx = cc.tap - ky * 3;
              tapb = (ky * PW + kx) * PS;
            }
            const int ab = tapb + cc.c * 64;
            f16x8 fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *(const f16x8*)(patch + aoff0[i] + ab);
            const f16* tB = slot + s * BN * 32 + frag_off;
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[j] = *(const f16x8*)(tB + j * 512);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
          }
          cc.next();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    tglob += nst;
  }
  if (is_loader) return;

  // ------------------------------------------------ sum the K slices (fixed order), then the epilogue
  float* red = (float*)smem;
  if constexpr (KW > 1) {
    constexpr int NF = MI * NI;
    // [KW - 1][WM][NF][64][4] floats over the ring + patch (every LDS reader has passed the last barrier)
    if (kw > 0) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          *(f32x4*)(red + ((((kw - 1) * WM + wm) * NF + i * NI + j) * 64 + lane) * 4) = acc[i][j];
    }
    __syncthreads();  // MFMA waves only: the loaders have ended
    if (kw > 0) return;
#pragma unroll
    for (int k = 1; k < KW; ++k)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] += *(const f32x4*)(red + ((((k - 1) * WM + wm) * NF + i * NI + j) * 64 + lane) * 4);
    red += (KW - 1) * WM * NF * 256;  // scratch of the GroupNorm partials lives behind the slice buffer
  }
  const int mlim = min(a.M, m0 + a.tile_rows);
  Epi::tile<MI, NI, WM, 1>(a, m0, m0 + wm * (MI * 16), n0, lc, lg, acc, wm, 0, red, mlim);
}

struct PcCfg {
  int mi, ni, wm, kw, ks, nbuf;
  const char* name;
  void (*fn)(const IgemmArgs);
};
#define PCFG(MI, NI, WM, KW, KS, NB) \
  {MI, NI, WM, KW, KS, NB, "p" #MI "x" #NI "w" #WM "k" #KW "s" #KS, pconv_kernel<MI, NI, WM, KW, KS, NB>}
const PcCfg kPc[] = {
    PCFG(4, 2, 4, 1, 2, 3),  // 256 x 32
    PCFG(2, 2, 4, 1, 2, 3),  // 128 x 32
    PCFG(1, 2, 4, 1, 2, 3),  //  64 x 32
    PCFG(4, 4, 4, 1, 2, 3),  // 256 x 64
    PCFG(2, 4, 4, 1, 2, 3),  // 128 x 64
    PCFG(1, 4, 4, 1, 2, 3),  //  64 x 64
    PCFG(2, 7, 4, 1, 2, 3),  // 128 x 112
    PCFG(1, 7, 4, 1, 2, 3),  //  64 x 112
    PCFG(3, 2, 4, 1, 2, 3),  // 192 x 32   (24-wide rows)
    PCFG(3, 4, 4, 1, 2, 3),  // 192 x 64
    PCFG(2, 2, 2, 2, 2, 3),  //  64 x 32, 2 K slices
    PCFG(2, 4, 2, 2, 2, 3),  //  64 x 64, 2 K slices
    PCFG(1, 2, 2, 2, 2, 3),  //  32 x 32
    PCFG(1, 4, 2, 2, 2, 3),  //  32 x 64
    PCFG(1, 2, 1, 4, 4, 3),  //  16 x 32, 4 K slices (4x4 level)
    PCFG(1, 4, 1, 4, 4, 3),  //  16 x 64
    PCFG(1, 1, 1, 4, 4, 3),  //  16 x 16
    PCFG(3, 2, 1, 4, 4, 3),  //  48 x 32
    PCFG(3, 4, 1, 4, 4, 3),  //  48 x 64
    PCFG(4, 1, 4, 1, 2, 3),  // 256 x 16  (N <= 16: output conv)
    PCFG(1, 1, 4, 1, 2, 3),  //  64 x 16
};
constexpr int kNumPc = sizeof(kPc) / sizeof(kPc[0]);
constexpr int kLdsBudget = 160 * 1024;

struct PcPlan {
  int cfg, tile_rows, pw, np, ps, cs, tpp_log2, tiles_m, tiles_n;
  size_t lds;
  double cost;
};

inline int cdivi(int a, int b) { return (a + b - 1) / b; }

// Geometry of configuration c for the launch `a` (ok == false: does not apply).
bool pc_plan(const PcCfg& c, const IgemmArgs& a, bool tiles_in_sample, int num_cus, PcPlan* p) {
  const int BM = c.mi * 16 * c.wm, BN = c.ni * 16;
  const int hw = a.Ho * a.Wo;
  if ((a.flags & UPK_F_GEGLU) && (c.ni % 4)) return false;
  int tr;
  if (a.ks == 3) {
    if (a.Wo > BM) return false;
    int R = BM / a.Wo;
    if (R > a.Ho) R = a.Ho;
    while (R > 1 && a.Ho % R) --R;
    tr = R * a.Wo;
    p->pw = (a.Wo % 16 == 0) ? a.Wo + 2 : a.Wo + 8;
    p->np = (R + 2) * p->pw;
  } else {
    tr = BM;
    if (tiles_in_sample) {
      tr = BM < hw ? BM : hw;
      while (tr > 1 && hw % tr) --tr;
    }
    p->pw = 1;
    p->np = BM;
  }
  if (tr * 2 <= BM && BM > 16) return false;  // a smaller tile configuration does the same work
  p->tile_rows = tr;
  p->tiles_m = (a.ks == 3 || tiles_in_sample) ? a.M / tr : cdivi(a.M, tr);
  if ((a.ks == 3 || tiles_in_sample) && a.M % tr) return false;
  p->tiles_n = cdivi(a.npad, BN);
  const int ring = c.nbuf * c.ks * BN * 64;
  const int fixed = ring + 2048;
  const int ctot = a.c1 + a.c2, capp = a.c3 + a.c4;
  const int cmax = ctot > capp ? ctot : capp;
  long budget = kLdsBudget - fixed;
  int cs_max = (int)((budget / p->np - 32) / 2) / 32 * 32;
  if (cs_max > 1024) cs_max = 1024;
  if (cs_max < 32) return false;
  const int nsl = cdivi(cmax, cs_max);
  int cs = cdivi(cdivi(cmax, nsl), 32) * 32;
  p->cs = cs;
  p->ps = 2 * cs + 32;
  int tl = 2;
  while ((1 << tl) < cs / 8) ++tl;
  p->tpp_log2 = tl;
  size_t patch = (size_t)p->np * p->ps;
  const size_t scratch = (size_t)2 * ctot * sizeof(float);
  if (a.gni_mode == 2 && patch < scratch) patch = scratch;
  const size_t kwred = c.kw > 1 ? (size_t)(c.kw - 1) * c.wm * c.mi * c.ni * 1024 + 4096 : 0;
  size_t lds = fixed + patch;
  if (lds < kwred + 8192) lds = kwred + 8192;
  if (lds > (size_t)kLdsBudget) return false;
  p->lds = lds;
  // rough cost: per-CU fill bytes (weights + patch, ~80 GB/s per CU) vs MFMA time, times the number of rounds
  const long wgs = (long)p->tiles_m * p->tiles_n;
  const double rounds = (double)cdivi((int)wgs, num_cus);
  const double K = (double)a.nchunks * 32.0;
  const double fill = (BN * K * 2.0 + (double)p->np * (ctot + capp) * 2.0) / 36.0;     // cycles
  const double mfma = (double)c.mi * c.ni * 16.0 * a.nchunks / c.kw * (BM / (double)tr > 1.3 ? 1.0 : 1.0);
  const double stage = (double)p->np * (ctot + capp) / 8.0 / 512.0 * (a.gni_mode ? 250.0 : 40.0);
  p->cost = rounds * ((fill > mfma ? fill : mfma) + stage + 6000.0);
  return true;
}

}  // namespace

extern "C" int upk_pconv_num_configs(void) { return kNumPc; }
extern "C" const char* upk_pconv_config_name(int cfg) { return (cfg >= 0 && cfg < kNumPc) ? kPc[cfg].name : "?"; }

namespace upkd {

// Launches `a` (prepared by conv_impl: everything but the tile decomposition) on the patch kernel.
// *handled == false: the shape is outside the kernel's domain, nothing was launched.
int pconv_run(upk_ctx* ctx, const upk_conv_desc* d, IgemmArgs& a, hipStream_t stream, bool launch, int* gn_fused,
              int* gn_nblk, bool* handled) {
  *handled = false;
  if (a.stride != 1 || a.ups || (a.flags & UPK_F_PAD_ASYM) || a.ln_u || a.vt) return UPK_OK;
  if (a.ks != 1 && a.ks != 3) return UPK_OK;
  const int hw = a.Ho * a.Wo;
  if (d->gni_mode) {
    if (d->gni_mode != 1 && d->gni_mode != 2) return upk_fail(ctx, UPK_EINVAL, "conv: gni_mode %d", d->gni_mode);
    if (!d->gni_gamma || !d->gni_beta || !d->gni_stats1 || d->gni_groups <= 0 || d->gni_groups > UPK_GN_GROUPS_MAX ||
        (a.c1 + a.c2) % d->gni_groups)
      return upk_fail(ctx, UPK_EINVAL, "conv: fused input GroupNorm needs gamma / beta / statistics and groups | channels");
    if (d->gni_mode == 2 && (d->gni_nblk1 <= 0 || d->gni_ld1 < a.c1 ||
                             (a.c2 > 0 && (!d->gni_stats2 || d->gni_nblk2 <= 0 || d->gni_ld2 < a.c2))))
      return upk_fail(ctx, UPK_EINVAL, "conv: channel partials of every source are needed for gni_mode 2");
    if (d->gni_mode == 1 && d->gni_nblk1 <= 0) return upk_fail(ctx, UPK_EINVAL, "conv: gni_nblk1");
    a.gni_mode = d->gni_mode;
    a.gni_silu = d->gni_silu;
    a.gni_groups = d->gni_groups;
    a.gni_cpg = (a.c1 + a.c2) / d->gni_groups;
    a.gni_eps = d->gni_eps;
    a.gni_gamma = d->gni_gamma;
    a.gni_beta = d->gni_beta;
    a.gni_s1 = d->gni_stats1;
    a.gni_s2 = d->gni_stats2;
    a.gni_nblk1 = d->gni_nblk1;
    a.gni_ld1 = d->gni_ld1;
    a.gni_nblk2 = d->gni_nblk2;
    a.gni_ld2 = d->gni_ld2;
  }
  const bool want_stats = d->gn_stats_ws && Epi::plain(a) && d->gn_groups > 0 && d->gn_groups <= UPK_GN_GROUPS_MAX &&
                          a.n_out % d->gn_groups == 0 && a.n_out <= 2048;
  const bool in_sample = a.gni_mode != 0 || want_stats;
  int best = -1;
  PcPlan bp;
  const int want = d->pc_cfg > 0 ? d->pc_cfg - 1 : -1;
  for (int c = 0; c < kNumPc; ++c) {
    if (want >= 0 && c != want) continue;
    PcPlan p;
    if (!pc_plan(kPc[c], a, in_sample, ctx->num_cus, &p)) continue;
    p.cfg = c;
    if (best < 0 || p.cost < bp.cost) {
      best = c;
      bp = p;
    }
  }
  if (best < 0) return UPK_OK;
  *handled = true;
  a.tile_rows = bp.tile_rows;
  a.p_pw = bp.pw;
  a.p_np = bp.np;
  a.p_ps = bp.ps;
  a.p_cs = bp.cs;
  a.p_tpp_log2 = bp.tpp_log2;
  a.tiles_m = bp.tiles_m;
  a.tiles_n = bp.tiles_n;
  a.partial = nullptr;
  const bool gn_cp = want_stats && hw % bp.tile_rows == 0 && hw / bp.tile_rows <= UPK_GN_MAX_CHUNKS;
  if (gn_cp) {
    a.gn_cp = d->gn_stats_ws;
    a.gn_nblk = hw / bp.tile_rows;
    a.gn_hw = hw;
  }
  if (gn_fused) *gn_fused = gn_cp ? 2 : 0;
  if (gn_nblk) *gn_nblk = gn_cp ? a.gn_nblk : 0;
  if (!launch) return UPK_OK;
  static bool attr_done[kNumPc] = {};
  if (!attr_done[best]) {
    UPK_HIP(ctx, hipFuncSetAttribute((const void*)kPc[best].fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
    attr_done[best] = true;
  }
  upk_prof_scope prof(ctx, UPK_CLS_IGEMM, stream);
  hipLaunchKernelGGL(kPc[best].fn, dim3(a.tiles_m * a.tiles_n), dim3(512), bp.lds, stream, a);
  return upk_check_launch(ctx, "pconv");
}

}  // namespace upkd

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
#include <type_traits>

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


// debug-only ablation bits (env UPK_ABLATE, dev builds -DUPK_DEV): which phase owns the time?
enum { PABL_NOEPI = 0x10000, PABL_NOGLOAD = 0x20000, PABL_NOSTAGE = 0x40000, PABL_NOMFMA = 0x80000, PABL_NOXF = 0x100000 };
#if defined(UPK_DEV)
#define PABL(f) ((a.flags & (f)) != 0)
// s_memtime stamps of block 0 (and the last block): consumer wave 0 -> slots 0.., loader wave 4 -> slots 16..
#define PSTAMP(i)                                                                                     \
  do {                                                                                                \
    if ((a.flags & 0x200000) && lane == 0 && (wave == 0 || wave == 4) &&                              \
        (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                                             \
      a.dbg[(blockIdx.x == 0 ? 0 : 32) + (wave == 0 ? 0 : 16) + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define PABL(f) (false)
#define PSTAMP(i) do { } while (0)
#endif

// One slab of the K loop: up to p_cs channels of ONE source tensor (x1, x2: the main ks x ks conv; x3, x4: the
// appended 1x1 segment), walked tap-major in 32-channel chunks and padded with dead chunks (zero weights) to whole
// stages.  Slabs are enumerated source by source (a slab never straddles a concat seam) by the HOST and travel in the
// kernel arguments: a slab's parameters are one scalar load, not live state (the kernel was spilling 70-100 SGPRs).
struct PcSlabE {
  const f16* base;  // source + first channel
  int ld;           // pixel stride of the source (elements)
  int cpsl;         // 32-channel chunks per tap
  int ntap;         // ks * ks, or 1 for the appended segment
  int nst;          // stages = ceil(ntap * cpsl / KS)
  int kcbase;       // first weight chunk (of tap 0)
  int ctab;         // channel of the concatenated input (GroupNorm table index), or -1: no transform
};
constexpr int kMaxSlabs = 60;
struct PcArgs {
  IgemmArgs a;
  int nslab;
  int pad_;
  PcSlabE slabs[kMaxSlabs];
};

// Position of a wave inside a slab's chunk sequence, advanced STEP chunks at a time without branches.
template <int STEP>
struct PcPos {
  int tap, c, dq, dr;
  __device__ __forceinline__ void start(int first, int cpsl) {
    tap = first / cpsl;
    c = first - tap * cpsl;
    dq = STEP / cpsl;
    dr = STEP - dq * cpsl;
  }
  __device__ __forceinline__ void step(int cpsl) {
    c += dr;
    tap += dq;
    const bool wrap = c >= cpsl;
    c = wrap ? c - cpsl : c;
    tap = wrap ? tap + 1 : tap;
  }
};

// 16-byte global load the compiler does not count (it would wait vmcnt(0) for an ordinary load and drain the
// weight DMAs in flight); completion is waited for by hand with pc_wait_x.
__device__ __forceinline__ void pc_load_x(f16x8& x, const f16* base, unsigned byte_off) {
  // (the base is wave-uniform by construction; readfirstlane makes that provable for the "s" constraint)
  const unsigned long long b = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  const unsigned long long bu = ((unsigned long long)hi << 32) | lo;
  // (s_nop 4: a VMEM instruction reading an SGPR a VALU instruction just wrote needs 5 wait states, and hipcc does
  // not pad inside an asm statement)
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(x) : "v"(byte_off), "s"(bu) : "memory");
}
template <int N>
__device__ __forceinline__ void pc_wait_x(f16x8& x) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(N) : "memory");
}
__device__ __forceinline__ void pc_wait_x4(f16x8& x0, f16x8& x1, f16x8& x2, f16x8& x3) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : "memory");
}

// Weight cursor of a loader wave, advanced incrementally (no divisions, no tap decode per chunk).  Every member is
// wave-uniform; the readfirstlane wrappers keep the compiler from parking them in VGPRs / scratch.
__device__ __forceinline__ int pc_u(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <int KS>
struct PcLoader {
  int slab, st, tap, kx, c, kc, ab;
  int cpsl, ntap, nst, kcstep;
  unsigned slot;  // ring slot of the next stage to issue
  int phase;      // this wave's first chunk of every stage
  __device__ __forceinline__ void enter(const PcArgs& pa, int ctr) {
    if (slab < pa.nslab) {
      const PcSlabE& e = pa.slabs[slab];
      cpsl = pc_u(e.cpsl);
      ntap = pc_u(e.ntap);
      nst = pc_u(e.nst);
      kc = pc_u(e.kcbase);
      kcstep = pc_u(pa.a.cpt - e.cpsl);
      ab = ntap == 9 ? 0 : ctr;
    } else {
      ntap = 0;
    }
    st = tap = kx = c = 0;
#pragma unroll 1
    for (int q = 0; q < phase; ++q) next_chunk(pa.a);
  }
  __device__ __forceinline__ void next_chunk(const IgemmArgs& a) {
    c = pc_u(c + 1);
    kc = pc_u(kc + 1);
    ab = pc_u(ab + 64);
    if (c == cpsl) {  // next tap
      c = 0;
      tap = pc_u(tap + 1);
      kc = pc_u(kc + kcstep);
      int tstep = 0;
      if (ntap == 9) {
        tstep = a.p_ps;
        kx = pc_u(kx + 1);
        if (kx == 3) {
          kx = 0;
          tstep = (a.p_pw - 2) * a.p_ps;
        }
      }
      ab = pc_u(ab + tstep - cpsl * 64);
    }
  }
  __device__ __forceinline__ void next_stage(const PcArgs& pa, int ctr, int nbuf) {
    slot = slot + 1 == (unsigned)nbuf ? 0 : slot + 1;
    st = pc_u(st + 1);
    if (st == nst) {
      slab = pc_u(slab + 1);
      enter(pa, ctr);
    }
  }
};

template <int MI, int NI, int WM, int KW, int KS, int NBUF, bool DB = true>
__global__ __launch_bounds__(512) void pconv_kernel(const PcArgs pa) {
  const IgemmArgs& a = pa.a;
  const int nslab = pa.nslab;
  // the kernel arguments of the prologue in ONE batch of scalar loads (read lazily they are a chain of ~10 dependent
  // scalar-cache misses before the first DMA, and inside the forward every one of them goes to memory)
  asm volatile("" ::"s"(a.p_xcd), "s"(a.tiles_m), "s"(a.tiles_n), "s"(a.tile_rows), "s"(a.p_pw), "s"(a.Ho), "s"(a.Wo), "s"(a.ks),
               "s"(a.zero), "s"(a.npad), "s"(pa.nslab), "s"(a.p_ps), "s"(a.p_cs), "s"(a.cpt), "s"(a.w), "s"(a.p_tpp_log2),
               "s"(a.p_np), "s"(a.HS), "s"(a.WS), "s"(a.gni_mode), "s"(a.c1), "s"(a.c2), "s"(a.M), "s"(a.p_T), "s"(a.p_tab),
               "s"(a.flags));
  asm volatile("" ::"s"(a.gni_gamma), "s"(a.gni_beta), "s"(a.gni_s1), "s"(a.gni_nblk1), "s"(a.gni_ld1), "s"(a.gni_groups),
               "s"(a.gni_cpg), "s"(a.gni_eps), "s"(a.gni_silu), "s"(a.x1), "s"(a.x2), "s"(a.ld1), "s"(a.ld2), "s"(a.sh_hw),
               "s"(a.sh_w));
  static_assert(WM * KW == 4, "4 MFMA waves");
  static_assert(KS % KW == 0, "every K slice gets the same number of chunks per stage");
  constexpr int BN = NI * 16;
  // weight DMAs of a stage = KS chunks x NI 16-row groups (1 KiB each), dealt to the 4 loader waves: CP chunks are
  // walked in parallel by G = 4 / CP waves each (wave lw: chunks lw / G + CP * i, groups lw % G + G * g), so a wave
  // advances its cursor KS / CP times per stage instead of KS times (the last group round is padded with dump DMAs).
  constexpr int CP = (NI == 1) ? 4 : 2;
  static_assert(KS % CP == 0, "KS must be a multiple of the chunks issued in parallel");
  constexpr int G = 4 / CP;
  constexpr int BGW = (NI + G - 1) / G;
  constexpr int P = (KS / CP) * BGW;  // DMAs per loader wave per stage
  constexpr int D = NBUF - 1;
  static_assert(D >= 2, "ring protocol needs 3 slots");
  static_assert(D * P <= 63, "vmcnt range");
  constexpr int STAGE = KS * BN * 32;  // halfs per ring slot
  constexpr int RING_BYTES = NBUF * STAGE * 2;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f16* const ring = (f16*)smem;
  f16* const dump = ring + NBUF * STAGE;
  float* const gstat = (float*)(smem + RING_BYTES + 1024);         // mean[32], rstd[32]
  double* const gsum = (double*)(smem + RING_BYTES + 1024 + 256);  // [64]
  float* const tab = (float*)(smem + RING_BYTES + 2048);           // [c1 + c2][2] GroupNorm scale, shift
  unsigned char* const patch0 = smem + RING_BYTES + 2048 + a.p_tab;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = wave >= 4;
  PSTAMP(0);
  const int tile = blockIdx.x;
  int tn, tm;
  if (a.p_xcd) {
    // XCD-aware order: workgroup b runs on XCD b % 8 (observed placement, used for speed only).  All M tiles that read
    // one N tile's weights sit on ONE XCD, so the weights cross the fabric once instead of once per XCD.
    const int i = tile >> 3;
    const int grp = i / a.tiles_m;
    tm = i - grp * a.tiles_m;
    tn = (tile & 7) + 8 * grp;
    if (tn >= a.tiles_n) return;
  } else {
    tn = tile / a.tiles_m;
    tm = tile - tn * a.tiles_m;
  }
  const int m0 = tm * a.tile_rows;
  const int n0 = tn * BN;
  const int hw = a.Ho * a.Wo;
  const int bsmp = m0 / hw;
  const int y0 = (m0 - bsmp * hw) / a.Wo;
  const bool k3 = a.ks == 3;
  const int PW = a.p_pw, PS = a.p_ps;
  const int T = a.p_T;  // total number of stages

  // ------------------------------------------------ loader: weight cursor + the first D stages
  const int lw = wave - 4;
  const f16* zsrc = a.zero + (lane & 3) * 8;
  const unsigned wstep = (unsigned)a.npad * 64u;  // bytes per weight chunk
  // ---- weight cursor of the loader waves, kept incrementally in scalars: a wave issues one instruction per four
  // cycles, so the ~50 instructions a stage may cost are the budget of this whole block.
  int* const tbl = (int*)(smem + RING_BYTES + 1024 + 768);  // [NBUF][KS] A byte offset of every chunk in the ring
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(lds_ptr)smem;  // LDS address of the ring
  const int ctr = (a.ks == 3) ? (a.p_pw + 1) * a.p_ps : 0;          // byte offset of the centre tap
  PcLoader<KS> L;
  L.slab = 0;
  L.slot = 0;
  unsigned long long wl[BGW];  // this lane's weight row (chunk 0) per owned 16-row group
  unsigned okm[BGW];           // ~0 / 0: the row exists
  int issued = 0;
  auto dma = [&](unsigned long long src, unsigned dst_lds) {
    if (!PABL(PABL_NOGLOAD))
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(__UINTPTR_TYPE__)dst_lds, 16, 0, 0);
  };
  auto issue_stage = [&]() {
    const unsigned sbase = lds0 + L.slot * (STAGE * 2) + (lw / G) * (BN * 64);  // this wave's first chunk of the stage
#pragma unroll
    for (int i = 0; i < KS / CP; ++i) {
      // a dead chunk (past the slab's last tap / past the last slab) gets zero weights: any finite A will do
      const bool live = L.tap < L.ntap;
      if (lw % G == 0) tbl[L.slot * KS + lw / G + CP * i] = live ? L.ab : ctr;
      if (live) {
        const unsigned koff = (unsigned)L.kc * wstep;
#pragma unroll
        for (int g = 0; g < BGW; ++g) {
          const int rg = lw % G + G * g;  // wave-uniform
          dma(wl[g] + (koff & okm[g]), (rg < NI) ? sbase + (CP * i * BN + rg * 16) * 64 : lds0 + RING_BYTES);
        }
      } else {
#pragma unroll
        for (int g = 0; g < BGW; ++g) {
          const int rg = lw % G + G * g;
          dma((unsigned long long)zsrc, (rg < NI) ? sbase + (CP * i * BN + rg * 16) * 64 : lds0 + RING_BYTES);
        }
      }
#pragma unroll
      for (int q = 0; q < CP; ++q) L.next_chunk(a);
    }
    L.next_stage(pa, ctr, NBUF);
  };
  if (is_loader) {
    const int r16 = lane >> 2;
    const int chd = (lane & 3) ^ ((-(lane >> 4)) & 3);  // source chunk of this lane (XOR swizzle on the source)
#pragma unroll
    for (int g = 0; g < BGW; ++g) {
      const int rg = lw % G + G * g;
      const int row = rg * 16 + r16;
      const bool ok = (rg < NI) && (n0 + row < a.npad);
      okm[g] = ok ? ~0u : 0u;
      wl[g] = ok ? (unsigned long long)(a.w + (long)(n0 + row) * 32 + chd * 8) : (unsigned long long)zsrc;
    }
    L.phase = lw / G;
    L.enter(pa, ctr);
    for (; issued < D; ++issued) issue_stage();  // (stages past the end are zero-page DMAs)
  }
  PSTAMP(1);

  // ------------------------------------------------ staging of a slab's patch
  // Two teams of 256 threads (MFMA waves, loader waves).  A thread owns ONE 8-channel vector position v of every
  // patch pixel it visits; piece p of a team visits pixel slot + p * nslots.  All address math is 32-bit.
  const int tl = a.p_tpp_log2;
  const int ltid = tid & 255;
  const int sv = ltid & ((1 << tl) - 1);
  const int sslot = ltid >> tl;
  const int nslots = 256 >> tl;
  const int npieces = (a.p_np + nslots - 1) / nslots;
  const int Rimg = a.tile_rows / a.Wo;  // image rows of the tile (3x3)
  const float inv_pw = 1.0f / (float)PW;
  const int rowbase = k3 ? (bsmp * a.HS + y0 - 1) * a.WS - 1 : m0;  // pixel index of patch (row 0, col 0)
  struct Piece {
    f16x8 x;
    int doff;  // LDS byte offset inside the patch, or -1: nothing to store
    bool val;  // loaded value (else zeros: padding)
  };
  // per-slab staging state of this thread
  float sc[8], sh[8];
  bool s_xf;
  auto slab_coeffs = [&](const PcSlabE& sl) {
    s_xf = sl.ctab >= 0 && !PABL(PABL_NOXF);
    if (s_xf) {
      const int ch = sl.ctab + sv * 8;  // channel of the concatenated input
      const float* t = tab + 2 * (ch < a.c1 + a.c2 - 7 ? ch : 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 q = *(const f32x4*)(t + 4 * j);
        sc[2 * j] = q[0];
        sh[2 * j] = q[1];
        sc[2 * j + 1] = q[2];
        sh[2 * j + 1] = q[3];
      }
    }
  };
  auto piece_issue = [&](const PcSlabE& sl, int p, Piece& pc) {
    const int pix = sslot + p * nslots;
    const bool vact = sv < sl.cpsl * 4;
    bool need = vact && pix < a.p_np, inimg;
    int gp;
    if (k3) {
      const int ry = (int)(((float)pix + 0.5f) * inv_pw);
      const int cx = pix - ry * PW;
      const int iy = y0 - 1 + ry, ix = cx - 1;
      inimg = (unsigned)iy < (unsigned)a.HS && (unsigned)ix < (unsigned)a.WS;
      need = need && cx < a.WS + 2;
      if (sl.ntap == 1) need = need && ry >= 1 && ry <= Rimg && cx >= 1 && cx <= a.WS;  // only the centre tap reads it
      gp = rowbase + ry * a.WS + cx;
    } else {
      inimg = pix < a.tile_rows && m0 + pix < a.M;
      gp = rowbase + pix;
    }
    pc.val = need && inimg;
    pc.doff = need ? pix * PS + sv * 16 : -1;
    const unsigned off = pc.val ? ((unsigned)gp * (unsigned)sl.ld + (unsigned)sv * 8u) * 2u : 0u;
    pc_load_x(pc.x, sl.base, off);
  };
  auto piece_store = [&](const Piece& pc, unsigned char* patch) {
    if (pc.doff < 0) return;
    f16x8 y = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pc.val) {
      y = pc.x;
      if (s_xf) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (float)pc.x[j] * sc[j] + sh[j];
        if (a.gni_silu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = f[j] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * f[j]));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = (f16)f[j];
      }
    }
    *(f16x8*)(patch + pc.doff) = y;
  };
  // pieces [p_from, npieces) of a slab by both teams (team m takes p_from + m, + 2, ..), 4 loads in flight
  auto stage_sync = [&](const PcSlabE& sl, int p_from, unsigned char* patch) {
    for (int p = p_from + (tid >> 8); p < npieces; p += 8) {
      Piece pc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) piece_issue(sl, p + 2 * u < npieces ? p + 2 * u : npieces, pc[u]);
      pc_wait_x4(pc[0].x, pc[1].x, pc[2].x, pc[3].x);
#pragma unroll
      for (int u = 0; u < 4; ++u) piece_store(pc[u], patch);
    }
  };

  PSTAMP(2);

  // ------------------------------------------------ GroupNorm statistics of the input (all waves)
  // Every fold issues ALL its loads before the first add: a runtime-trip-count "load, accumulate" loop is compiled into
  // that many dependent L2 round trips, and this prologue is on the critical path of every launch.
  if (a.gni_mode) {
    const int groups = a.gni_groups, cpg = a.gni_cpg;
    const int ctot = a.c1 + a.c2;
    // gamma / beta of this thread's channels: requested before the statistics
    float pg[4], pb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ch = tid + k * 512;
      pg[k] = ch < ctot ? a.gni_gamma[ch] : 0.f;
      pb[k] = ch < ctot ? a.gni_beta[ch] : 0.f;
    }
    if (a.gni_mode == 2) {
      float* chs = (float*)patch0;  // [2][ctot] channel sums; the patch is staged after the last read
      const float* w1 = a.gni_s1 + (long)bsmp * a.gni_nblk1 * 2 * a.gni_ld1;
      const float* w2 = a.gni_s2 ? a.gni_s2 + (long)bsmp * a.gni_nblk2 * 2 * a.gni_ld2 : nullptr;
      constexpr int EU = 2, KU = 8;  // entries x row blocks in flight per thread
      for (int i0 = tid; i0 < 2 * ctot; i0 += 512 * EU) {
        const float* src[EU];
        int ldk[EU], nb[EU];
        float acc[EU];
#pragma unroll
        for (int e = 0; e < EU; ++e) {
          const int idx = i0 + e * 512;
          const bool ok = idx < 2 * ctot;
          const int which = idx >= ctot ? 1 : 0;
          const int ch = ok ? idx - which * ctot : 0;
          const bool second = ch >= a.c1;
          const int ld = second ? a.gni_ld2 : a.gni_ld1;
          src[e] = (second ? w2 + (ch - a.c1) : w1 + ch) + which * ld;
          ldk[e] = 2 * ld;
          nb[e] = ok ? (second ? a.gni_nblk2 : a.gni_nblk1) : 0;
          acc[e] = 0.f;
        }
        const int nbmax = max(a.gni_nblk1, a.gni_nblk2);
        for (int k0 = 0; k0 < nbmax; k0 += KU) {
          float v[EU][KU];
#pragma unroll
          for (int e = 0; e < EU; ++e)
#pragma unroll
            for (int k = 0; k < KU; ++k) v[e][k] = (k0 + k < nb[e]) ? src[e][(long)(k0 + k) * ldk[e]] : 0.f;
#pragma unroll
          for (int e = 0; e < EU; ++e)
#pragma unroll
            for (int k = 0; k < KU; ++k) acc[e] += v[e][k];
        }
#pragma unroll
        for (int e = 0; e < EU; ++e)
          if (i0 + e * 512 < 2 * ctot) chs[i0 + e * 512] = acc[e];
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
      // per-(chunk, group) partials [nblk1 <= 32][groups][2]: 4 threads per (group, sum | sumsq), 8 loads each
      double* part = (double*)patch0;  // [groups * 2][4]
      const int q = tid >> 2, sub = tid & 3;
      if (q < groups * 2) {
        const float* w = a.gni_s1 + (long)bsmp * a.gni_nblk1 * groups * 2 + q;
        float v[UPK_GN_MAX_CHUNKS / 4];
#pragma unroll
        for (int k = 0; k < UPK_GN_MAX_CHUNKS / 4; ++k) {
          const int idx = sub + 4 * k;
          v[k] = idx < a.gni_nblk1 ? w[(long)idx * groups * 2] : 0.f;
        }
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < UPK_GN_MAX_CHUNKS / 4; ++k) acc += (double)v[k];
        part[tid] = acc;
      }
      pc_bar();
      if (tid < groups * 2) gsum[tid] = ((part[tid * 4] + part[tid * 4 + 1]) + part[tid * 4 + 2]) + part[tid * 4 + 3];
      pc_bar();
    }
    if (tid < groups) {
      const double n = (double)hw * cpg;
      const double mean = gsum[tid * 2] / n;
      double var = gsum[tid * 2 + 1] / n - mean * mean;
      if (var < 0.0) var = 0.0;
      gstat[tid] = (float)mean;
      gstat[32 + tid] = rsqrtf((float)var + a.gni_eps);
    }
    pc_bar();
    // per-channel scale / shift of the whole (concatenated) input, read by the staging passes of every slab
    const float inv_cpg = 1.0f / (float)cpg;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ch = tid + k * 512;
      if (ch < ctot) {
        const int g = (int)(((float)ch + 0.5f) * inv_cpg);
        const float s = gstat[32 + g] * pg[k];
        tab[2 * ch] = s;
        tab[2 * ch + 1] = pb[k] - gstat[g] * s;
      }
    }
    pc_bar();
  }
  PSTAMP(3);

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
  constexpr int CH = KS / KW;              // chunks per MFMA wave per stage

  // ------------------------------------------------ slab 0 is staged by everybody
  if (!PABL(PABL_NOSTAGE)) {
    slab_coeffs(pa.slabs[0]);
    stage_sync(pa.slabs[0], 0, patch0);
  }
  PSTAMP(4);
  if (is_loader) pc_wait_vmcnt<0>();  // (the staging loads were waited for with vmcnt(0): the first D stages landed too)
  PSTAMP(5);
  pc_bar();
  PSTAMP(6);

  // ------------------------------------------------ slabs
  // Ring protocol: when the barrier that ends stage g falls, stage g + 2 has landed (the loaders waited for it), so
  // the MFMA waves fetch the fragments of a stage's first chunk while they are still multiplying the last chunk of
  // the stage before: fragment reads and MFMAs overlap inside ONE wave (register double buffer).
  int tglob = 0;
  for (int slab_i = 0; slab_i < nslab; ++slab_i) {
    const int sl_nst = pa.slabs[slab_i].nst;
    unsigned char* const patch = patch0;  // (one buffer: every slab is staged by all waves between two K loops)
    if (is_loader) {
      for (int t = 0; t < sl_nst; ++t) {
        issue_stage();
        ++issued;
        pc_wait_vmcnt<(D - 2) * P>();  // stage g + 2 has landed: D - 2 younger stages may stay in flight
        __builtin_amdgcn_s_barrier();
      }
    } else {
      // K loop of this slab: per chunk MI address adds, MI + NI fragment reads, MI * NI MFMAs.  The patch offsets come
      // from `tbl` (one small LDS read per stage, a stage ahead); fragments are double-buffered in registers.
      f16x8 fa[DB ? 2 : 1][MI], fb[DB ? 2 : 1][NI];
      int tcur[CH], tnxt[CH];
      auto tbl_load = [&](int (&tt)[CH], int slot) {
#pragma unroll
        for (int j = 0; j < CH; ++j) tt[j] = tbl[slot * KS + kw + KW * j];
      };
      auto frag_load = [&](f16x8(&xa)[MI], f16x8(&xb)[NI], int ab, const f16* tB) {
#pragma unroll
        for (int i = 0; i < MI; ++i) xa[i] = *(const f16x8*)(patch + aoff0[i] + ab);
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) xb[jj] = *(const f16x8*)(tB + jj * 512);
      };
      auto mma = [&](const f16x8(&xa)[MI], const f16x8(&xb)[NI]) {
        if (!PABL(PABL_NOMFMA)) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jj = 0; jj < NI; ++jj)
              acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[jj], xa[i], acc[i][jj], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(xa[i]));
#pragma unroll
          for (int jj = 0; jj < NI; ++jj) asm volatile("" ::"v"(xb[jj]));
        }
      };
      const f16* const bbase = ring + kw * (BN * 32) + frag_off;
      int slot = tglob % NBUF;
      // one stage; PAR = parity of the register buffer that holds its first chunk
      auto stage = [&](auto par, int t) {
        constexpr int PAR = decltype(par)::value;
        const f16* sb = bbase + slot * STAGE;
        const int nslot = slot + 1 == NBUF ? 0 : slot + 1;
        const bool more = t + 1 < sl_nst;
        if (more) tbl_load(tnxt, nslot);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          if constexpr (DB) {
            const int cur = (PAR * CH + j) & 1;
            if (j + 1 < CH) {
              if (cur == 0) frag_load(fa[1], fb[1], tcur[j + 1 < CH ? j + 1 : 0], sb + (j + 1) * (KW * BN * 32));
              else frag_load(fa[0], fb[0], tcur[j + 1 < CH ? j + 1 : 0], sb + (j + 1) * (KW * BN * 32));
            } else if (more) {
              const f16* nb = bbase + nslot * STAGE;
              if (cur == 0) frag_load(fa[1], fb[1], tnxt[0], nb);
              else frag_load(fa[0], fb[0], tnxt[0], nb);
            }
            if (cur == 0) mma(fa[0], fb[0]);
            else mma(fa[1], fb[1]);
          } else {  // big register tiles: one fragment set, the MFMAs start as the first fragments land
            frag_load(fa[0], fb[0], tcur[j], sb + j * (KW * BN * 32));
            mma(fa[0], fb[0]);
          }
        }
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < CH; ++j) tcur[j] = tnxt[j];
        slot = nslot;
      };
      tbl_load(tcur, slot);
      if constexpr (DB) frag_load(fa[0], fb[0], tcur[0], bbase + slot * STAGE);
      int t = 0;
      for (; t + 1 < sl_nst; t += 2) {
        stage(std::integral_constant<int, 0>{}, t);
        stage(std::integral_constant<int, 1>{}, t + 1);
      }
      if (t < sl_nst) stage(std::integral_constant<int, 0>{}, t);
    }
    tglob += sl_nst;
    if (slab_i == 0) PSTAMP(7);
    if (slab_i + 1 < nslab && !PABL(PABL_NOSTAGE)) {  // the next slab's patch, by everybody (the MFMA pipe idles)
      slab_coeffs(pa.slabs[slab_i + 1]);
      stage_sync(pa.slabs[slab_i + 1], 0, patch0);
      pc_bar();
    }
  }
  PSTAMP(8);
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
  if (PABL(PABL_NOEPI)) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 123.456f) ((float*)a.y)[0] = t;
    return;
  }
  PSTAMP(9);
  const int mlim = min(a.M, m0 + a.tile_rows);
  Epi::tile<MI, NI, WM, 1>(a, m0, m0 + wm * (MI * 16), n0, lc, lg, acc, wm, 0, red, mlim);
#if defined(UPK_DEV)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PSTAMP(10);
#endif
}

struct PcCfg {
  int mi, ni, wm, kw, ks, nbuf;
  const char* name;
  void (*fn)(const PcArgs);
};
#define PCFG(MI, NI, WM, KW, KS, NB) \
  {MI, NI, WM, KW, KS, NB, "p" #MI "x" #NI "w" #WM "k" #KW "s" #KS "r" #NB, pconv_kernel<MI, NI, WM, KW, KS, NB>}
#define PCFG1(MI, NI, WM, KW, KS, NB) \
  {MI, NI, WM, KW, KS, NB, "p" #MI "x" #NI "w" #WM "k" #KW "s" #KS "r" #NB "f1", pconv_kernel<MI, NI, WM, KW, KS, NB, false>}
// (MI, NI, WM, KW, KS, NBUF): tile = (16 MI WM) rows x (16 NI) columns, KW K slices, KS chunks per stage, ring slots.
// Ring depth: weights arrive ~1 us after their DMA is issued, so a CU needs tens of KB in flight to fill at its
// ~80 GB/s — 5-8 slots of 8-16 KB; the 3-slot forms are kept for tiles whose patch needs the LDS.
const PcCfg kPc[] = {
    PCFG(4, 2, 4, 1, 4, 6), PCFG(4, 2, 4, 1, 2, 4),                                                  // 256 rows
    PCFG(2, 2, 4, 1, 4, 8), PCFG(2, 4, 4, 1, 4, 5), PCFG(2, 4, 4, 1, 2, 6),                          // 128 rows
    PCFG(1, 2, 4, 1, 4, 8), PCFG(1, 4, 4, 1, 4, 5), PCFG(1, 7, 4, 1, 2, 5), PCFG(1, 7, 4, 1, 2, 4),  //  64 rows
    PCFG(2, 2, 2, 2, 4, 8), PCFG(2, 4, 2, 2, 4, 5), PCFG(2, 7, 2, 2, 2, 5), PCFG(4, 2, 1, 4, 4, 8),  //  64 rows, K slices
    PCFG(4, 4, 1, 4, 4, 5), PCFG1(4, 7, 1, 4, 4, 4), PCFG1(4, 7, 1, 4, 4, 3), PCFG1(4, 4, 1, 4, 4, 5), PCFG1(4, 4, 1, 4, 8, 4),
    PCFG(1, 2, 2, 2, 4, 8), PCFG(1, 4, 2, 2, 4, 5), PCFG(2, 2, 1, 4, 4, 8), PCFG(2, 4, 1, 4, 4, 5),  //  32 rows
    PCFG(1, 2, 1, 4, 8, 5), PCFG(1, 2, 1, 4, 4, 8), PCFG(1, 4, 1, 4, 4, 5), PCFG(1, 1, 1, 4, 8, 8),  //  16 rows, 4 K slices
    PCFG(3, 2, 4, 1, 4, 8), PCFG(3, 2, 1, 4, 8, 5), PCFG(3, 2, 1, 4, 4, 8),                          // 192 / 48 rows (24-wide images)
    PCFG(4, 1, 4, 1, 4, 8), PCFG(1, 1, 4, 1, 4, 8),                                                  // N <= 16 (output conv)
};
constexpr int kNumPc = sizeof(kPc) / sizeof(kPc[0]);
constexpr int kLdsBudget = 160 * 1024;

struct PcPlan {
  int cfg, tile_rows, pw, np, ps, cs, tpp_log2, tiles_m, tiles_n, T, tab, xcd;
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
  const int ctot = a.c1 + a.c2, capp = a.c3 + a.c4;
  const int tabb = a.gni_mode ? (ctot * 8 + 255) / 256 * 256 : 0;
  const int fixed = ring + 2048 + tabb;
  int cmax = a.c1;
  if (a.c2 > cmax) cmax = a.c2;
  if (a.c3 > cmax) cmax = a.c3;
  if (a.c4 > cmax) cmax = a.c4;
  long budget = kLdsBudget - fixed;
  int cs_max = (int)((budget / p->np - 32) / 2) / 32 * 32;
  if (cs_max > 1024) cs_max = 1024;
  if (cs_max < 32) return false;
  int cs_want = 1024;  // as few slabs as fit (every slab boundary is a staging pass with the MFMA pipe idle)
  if (const char* e = getenv("UPK_PC_CS")) cs_want = atoi(e);
  if (cs_want < 32 * c.kw) cs_want = 32 * c.kw;
  if (cs_max > cs_want) cs_max = cs_want;
  const int nsl = cdivi(cmax, cs_max);
  int cs = cdivi(cdivi(cmax, nsl), 32) * 32;
  p->cs = cs;
  p->ps = 2 * cs + 32;
  p->tab = tabb;
  int tl = 2;
  while ((1 << tl) < cs / 8) ++tl;
  if (tl > 8) return false;
  p->tpp_log2 = tl;
  // stages: slab by slab over the four sources
  int T = 0, nsl_tot = 0;
  const int segs[4] = {a.c1, a.c2, a.c3, a.c4};
  for (int sg = 0; sg < 4; ++sg)
    for (int c0 = 0; c0 < segs[sg]; c0 += cs) {
      const int cpsl = ((segs[sg] - c0 < cs) ? segs[sg] - c0 : cs) / 32;
      T += cdivi((sg < 2 ? a.ks * a.ks : 1) * cpsl, c.ks);
      ++nsl_tot;
    }
  if (nsl_tot > kMaxSlabs) return false;
  p->T = T;
  p->xcd = p->tiles_n >= 6 ? 1 : 0;
  if (const char* e = getenv("UPK_PC_XCD")) p->xcd = atoi(e);
  size_t patch = (size_t)p->np * p->ps;
  const size_t scratch = (size_t)2 * ctot * sizeof(float);
  if (a.gni_mode == 2 && patch < scratch) patch = scratch;
  if (a.gni_mode == 1 && patch < 2048) patch = 2048;
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
    if (d->gni_nblk1 > UPK_GN_MAX_CHUNKS || (d->gni_mode == 2 && a.c2 > 0 && d->gni_nblk2 > UPK_GN_MAX_CHUNKS))
      return upk_fail(ctx, UPK_EINVAL, "conv: at most %d partial blocks per sample", UPK_GN_MAX_CHUNKS);
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
  a.p_T = bp.T;
  a.p_xcd = bp.xcd;
  a.p_tab = bp.tab;
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
  const int nwg = bp.xcd ? 8 * a.tiles_m * ((a.tiles_n + 7) / 8) : a.tiles_m * a.tiles_n;
  PcArgs pa;
  pa.a = a;
  pa.nslab = 0;
  pa.pad_ = 0;
  {
    const f16* srcs[4] = {a.x1, a.x2, a.x3, a.x4};
    const int segs[4] = {a.c1, a.c2, a.c3, a.c4};
    const int lds[4] = {a.ld1, a.ld2, a.ld3, a.ld4};
    for (int sg = 0; sg < 4; ++sg)
      for (int c0 = 0; c0 < segs[sg]; c0 += bp.cs) {
        PcSlabE& e = pa.slabs[pa.nslab++];
        const bool app = sg >= 2;
        e.base = srcs[sg] + c0;
        e.ld = lds[sg];
        e.cpsl = ((segs[sg] - c0 < bp.cs) ? segs[sg] - c0 : bp.cs) / 32;
        e.ntap = app ? 1 : a.ks * a.ks;
        e.nst = cdivi(e.ntap * e.cpsl, kPc[best].ks);
        e.kcbase = (app ? a.nchunks_main : 0) + (((sg & 1) ? segs[sg - 1] : 0) + c0) / 32;
        e.ctab = (!app && a.gni_mode) ? ((sg & 1) ? a.c1 : 0) + c0 : -1;
      }
  }
  hipLaunchKernelGGL(kPc[best].fn, dim3(nwg), dim3(512), bp.lds, stream, pa);
  return upk_check_launch(ctx, "pconv");
}

}  // namespace upkd

// Fused feed-forward tail of a SpatialTransformer block (attention.py:42-64, 215, 259-261), one launch:
//
//     h   = GEGLU(LayerNorm(t2))                      [M, 4C]    norm3 -> ff.net.0 (value | gate, erf GELU)
//     out = x_in + (P F2) h + P t2 + (P b2 + bp)      [M, C]     ff.net.2, + t2, proj_out, + x_in  (one GEMM over [h | t2])
//
// A workgroup owns BM rows.  t2's tile is staged once in LDS (LDS-DMA), the 4C-wide hidden activation h is produced
// pass by pass (8 waves x 2 fragments of 16 columns, astat.hip's scheme: weights straight from L2 into a register
// ring, no barrier in the K loops) and written to LDS as the A operand of the second GEMM — it never exists in HBM:
// 14.7 MB written and read back per block at the 32x32 level, and one launch less.  The weight stream runs on across
// the seam (the ring's refills go from the last GEGLU pass straight into the second GEMM's first chunks); the only
// workgroup barriers are the one behind the tile DMA and the one between the two GEMMs.
//
// The price is the weight traffic per CU: every workgroup streams BOTH weights in full ((8 + 5) C^2 fp16), where the
// two-launch form tiles N.  It pays where M / BM covers the chip (the 32x32 level: M = 8192), not at the deeper levels.
#include "igemm_common.h"

namespace upkd {
namespace {

constexpr int ML_NW = 8;

struct MlpArgs {
  const f16* x;      // t2 [M, ldx] (un-normalised residual stream)
  const f16* w1;     // packed GEGLU weight [nch1][n1][32], LayerNorm affine folded in (W * gamma)
  const f16* w2;     // packed [nh + nch1][n2pad][32]: K order [h | t2]
  const f16* zero;
  const float* b1;   // [n1] packed order (b + W beta)
  const float* u1;   // [n1] column sums of the fp16-rounded W * gamma
  const float* b2;   // [n2pad]
  const f16* res;    // x_in [M, ldr]
  f16* y;            // out [M, ldy]
  float* gn_cp;      // per-(M tile, channel) partials [B][nblk][2][n2pad] for the GroupNorm that reads `out`, or nullptr
  int ldx, ldr, ldy, M, nch1, n1, n2, n2pad, nh;
  int gn_nblk, gn_hw;
  float ln_inv_dim, ln_eps;
  unsigned long long* dbg;  // dev: s_memtime stamps (UPK_TIMELINE builds)
};
#define ML_PIN(v) asm volatile("" ::"s"(v))

template <int MI, int PF>
__global__ __launch_bounds__(512) void mlp_kernel(const MlpArgs s) {
  constexpr int NW = ML_NW, NI = 2;
  constexpr int BM = MI * 16;
  constexpr int PW = NW * NI * 16;  // packed GEGLU columns per pass (= 128 hidden columns)
  extern __shared__ __attribute__((aligned(16))) f16 smem[];
  ML_PIN(s.x); ML_PIN(s.w1); ML_PIN(s.w2); ML_PIN(s.zero); ML_PIN(s.b1); ML_PIN(s.u1); ML_PIN(s.b2); ML_PIN(s.res);
  ML_PIN(s.y); ML_PIN(s.gn_cp); ML_PIN(s.ldx); ML_PIN(s.ldr); ML_PIN(s.ldy); ML_PIN(s.M); ML_PIN(s.nch1); ML_PIN(s.n1);
  ML_PIN(s.n2); ML_PIN(s.n2pad); ML_PIN(s.nh);

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
  const int nch1 = s.nch1, nh = s.nh, nch2 = nh + nch1;
  // LDS: chunks [0, nh) = h, chunks [nh, nh + nch1) = t2  (the second GEMM's K order), then the row statistics
  f16* const hT = smem;
  f16* const xT = smem + nh * BM * 32;
  float* const st = (float*)(smem + nch2 * BM * 32);  // [BM][2]
  float* const bl = st + BM * 2;                      // [2][n1]: GEGLU bias, LayerNorm column sums (packed order)

  // ---- 1. t2 tile
  {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int r16 = lane >> 2;
    const int chd = (lane & 3) ^ ((-(lane >> 4)) & 3);
    const f16* zsrc = s.zero + (lane & 3) * 8;
    for (int idx = wave; idx < nch1 * MI; idx += NW) {
      const int kc = idx / MI, rg = idx - kc * MI;
      const int m = m0 + rg * 16 + r16;
      const f16* src = m < s.M ? s.x + (long)m * s.ldx + kc * 32 + chd * 8 : zsrc;
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(xT + (kc * BM + rg * 16) * 32), 16, 0, 0);
    }
    // the GEGLU epilogue's operands for every pass, into LDS with the tile: read from global memory per pass they put a
    // vmcnt(0) in front of each epilogue (the compiler cannot count across the K loop) — a drain of the 14 weight
    // fragments just requested for the next pass, ~2k cycles per pass in the in-kernel stamps
    const int nkb = s.n1 >> 8;  // 1 KiB pieces per array (n1 % 256 == 0)
    for (int idx = wave; idx < 2 * nkb; idx += NW) {
      const float* src = (idx < nkb ? s.b1 + idx * 256 : s.u1 + (idx - nkb) * 256) + lane * 4;
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(bl + idx * 256), 16, 0, 0);
    }
  }

  // ---- 2. weight streams.  GEGLU: wave w owns, per pass, value columns blk*64 + q*16 .. +15 and their gates 32
  // columns on (blk = pass * 4 + w / 2, q = w & 1).  Second GEMM: waves 0 .. n2pad / 32 - 1 own 32 columns each.
  const int col1 = (wave >> 1) * 64 + (wave & 1) * 16;
  const int npass = (s.n1 + PW - 1) / PW;
  const int p1w = min(npass, (s.n1 - col1 + PW - 1) / PW);  // passes with columns for this wave
  const int col2 = wave * 32;
  const bool act2 = col2 < s.n2pad;
  const unsigned ks1 = (unsigned)s.n1 * 64u, ks2 = (unsigned)s.n2pad * 64u;
  const unsigned loff = (unsigned)(lc * 32 + lg * 8) * 2u;
  unsigned voff1[PF], voff2[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) voff1[u] = loff + (unsigned)u * ks1, voff2[u] = loff + (unsigned)u * ks2;
  const char* wb1 = (const char*)s.w1;
  const char* wb2 = (const char*)s.w2 + (size_t)col2 * 64;
  auto base1 = [&](int p, int kc) -> const char* {
    return wb1 + (size_t)(((unsigned)kc * (unsigned)s.n1 + (unsigned)(col1 + p * PW)) * 64u);
  };
  f16x8 ring[PF][NI];
  if (p1w > 0) {
    const char* b0 = base1(0, 0);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      ring[u][0] = *(const f16x8*)(b0 + voff1[u]);
      ring[u][1] = *(const f16x8*)(b0 + 2048 + voff1[u]);
    }
  } else if (act2) {  // (no GEGLU columns for this wave: its ring starts in the second GEMM)
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      ring[u][0] = *(const f16x8*)(wb2 + voff2[u]);
      ring[u][1] = *(const f16x8*)(wb2 + 1024 + voff2[u]);
    }
  }
  // operands of the second GEMM's epilogue (requested with the ring: one round trip)
  f32x4 b2v[NI];
  f16x4 rr[MI][NI];
  if (act2) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = col2 + j * 16 + lg * 4;
      b2v[j] = *(const f32x4*)(s.b2 + (n < s.n2pad ? n : 0));
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const unsigned m = (unsigned)min(m0 + i * 16 + lc, s.M - 1);
        rr[i][j] = *(const f16x4*)(s.res + m * (unsigned)s.ldr + (unsigned)(n < s.n2 ? n : 0));
      }
    }
  }
  STAMP(1);
  __syncthreads();  // (t2 tile landed)
  STAMP(2);

  // ---- 3. LayerNorm statistics of the tile's rows (from the resident tile)
  {
    constexpr int LPR = 512 / BM;
    const int row = tid / LPR, part = tid - row * LPR;
    float s1 = 0.f, s2 = 0.f;
    const f16x2 one2 = {(f16)1.f, (f16)1.f};
    for (int q = part; q < nch1 * 4; q += LPR) {
      const f16x8 v = *(const f16x8*)(xT + ((q >> 2) * BM + row) * 32 + (q & 3) * 8);
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

  STAMP(3);
  const unsigned la = (unsigned)(lc * 32 + lds_swz(lc, lg) * 8) * 2u;
  const char* const xs = (const char*)xT;
  const char* const hs = (const char*)hT;

  // ---- 4. GEGLU passes -> h tile in LDS
  for (int p = 0; p < p1w; ++p) {
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kc0 = 0; kc0 < nch1; kc0 += PF) {
      const bool wrap = kc0 + PF >= nch1;       // the refills of this block belong to the next pass ...
      const bool seam = wrap && p + 1 >= p1w;   // ... or, behind the last pass, to the second GEMM's first chunks
      const char* sb0 = seam ? (act2 ? wb2 : wb1) : base1(wrap ? p + 1 : p, wrap ? 0 : kc0 + PF);
      const char* sb1 = seam ? (act2 ? wb2 + 1024 : wb1) : sb0 + 2048;
      const char* ldsA = xs + la + (unsigned)(kc0 * BM) * 64u;
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        f16x8 fc[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) fc[i] = *(const f16x8*)(ldsA + (u * BM + i * 16) * 64);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[u][j], fc[i], acc[i][j], 0, 0, 0);
        const unsigned vo = seam ? (act2 ? voff2[u] : loff) : voff1[u];
        ring[u][0] = *(const f16x8*)(sb0 + vo);
        ring[u][1] = *(const f16x8*)(sb1 + vo);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(4 + 2 * p);
    // epilogue: folded LayerNorm, bias, v * gelu(g) -> fp16 -> h tile (chunk = hidden column / 32, XOR-swizzled 16-byte
    // pieces like every A tile: this lane's 4 columns are half a piece)
    const int n = col1 + p * PW + lg * 4;            // packed value column
    const f32x4 bv0 = *(const f32x4*)(bl + n), bv1 = *(const f32x4*)(bl + n + 32);
    const f32x4 lu0 = *(const f32x4*)(bl + s.n1 + n), lu1 = *(const f32x4*)(bl + s.n1 + n + 32);
    const int oc = (n >> 6) * 32 + (n & 31);         // hidden column
    const int hk = oc >> 5, hq = (oc & 31) >> 3, hh = oc & 7;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const f32x2 mr = *(const f32x2*)(st + 2 * (i * 16 + lc));
      const f32x4 v = (acc[i][0] - mr[0] * lu0) * mr[1] + bv0;
      const f32x4 g = (acc[i][1] - mr[0] * lu1) * mr[1] + bv1;
      f16x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (f16)upk_geglu_mul(v[k], g[k]);
      const int row = i * 16 + lc;
      *(f16x4*)(hT + (hk * BM + row) * 32 + lds_swz(lc, hq) * 8 + hh) = o;
    }
    STAMP(5 + 2 * p);
  }
  STAMP(20);
  __syncthreads();  // (h complete)
  STAMP(21);

  // ---- 5. second GEMM over [h | t2] (LDS chunk order = its K order), epilogue: bias, residual, GroupNorm partials
  if (!act2) return;
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int kc0 = 0; kc0 < nch2; kc0 += PF) {
    const bool dead = kc0 + PF >= nch2;
    const char* sb0 = dead ? wb2 : wb2 + (size_t)((unsigned)(kc0 + PF) * ks2);
    const char* ldsA = hs + la + (unsigned)(kc0 * BM) * 64u;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      f16x8 fc[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fc[i] = *(const f16x8*)(ldsA + (u * BM + i * 16) * 64);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[u][j], fc[i], acc[i][j], 0, 0, 0);
      const unsigned vo = dead ? loff : voff2[u];
      ring[u][0] = *(const f16x8*)(sb0 + vo);
      ring[u][1] = *(const f16x8*)(sb0 + (dead ? 0 : 1024) + vo);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  STAMP(22);
  f32x4 cs[NI], cq[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) cs[j] = cq[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + i * 16 + lc;
    const bool ok = m < s.M;
    f16* yrow = s.y + (unsigned)(ok ? m : 0) * (unsigned)s.ldy;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = col2 + j * 16 + lg * 4;
      const f32x4 v = acc[i][j] + b2v[j];
      f16x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (f16)(v[k] + (float)rr[i][j][k]);
      if (ok && n < s.n2) *(f16x4*)(yrow + n) = o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float f = ok ? (float)o[k] : 0.f;
        cs[j][k] += f;
        cq[j][k] += f * f;
      }
    }
  }
  if (s.gn_cp) {  // (a wave owns its columns for all BM rows: the 16 rows of a fragment by rotate-add, no LDS)
    const int b = m0 / s.gn_hw;
    const int blk = (m0 - b * s.gn_hw) / BM;
    float* dst = s.gn_cp + (long)((b * s.gn_nblk + blk) * 2) * s.n2pad;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = col2 + j * 16 + lg * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        cs[j][k] = Epi::row_sum16(cs[j][k]);
        cq[j][k] = Epi::row_sum16(cq[j][k]);
      }
      if (lc == 0 && n < s.n2pad) {
        *(f32x4*)(dst + n) = cs[j];
        *(f32x4*)(dst + s.n2pad + n) = cq[j];
      }
    }
  }
#ifdef UPK_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(23);
#endif
#undef STAMP
}

}  // namespace
}  // namespace upkd

using namespace upkd;

extern "C" int upk_geglu_mlp_supported(upk_ctx* ctx, const upk_mlp_desc* d) {
  if (!ctx || !d) return 0;
  if (d->c <= 0 || (d->c & 31) || d->inner != 4 * d->c || (d->inner & 31)) return 0;
  const int nch1 = d->c / 32, nh = d->inner / 32;
  if (nch1 % 7 || (nch1 + nh) % 7) return 0;   // (ring depth 7: the 7 * 32 channel family)
  if (d->n_pad > ML_NW * 32 || (d->n_pad & 31)) return 0;
  if ((d->ldx & 7) || (d->ld_res & 3) || (d->ldy & 3) || (d->n_out & 3)) return 0;
  const int bm = d->rows_per_wg > 0 ? d->rows_per_wg : 64;
  if (bm != 32 && bm != 64) return 0;
  if ((2 * d->inner) & 255) return 0;
  if ((size_t)(nch1 + nh) * bm * 64 + (size_t)bm * 8 + (size_t)d->inner * 16 > 160 * 1024) return 0;
  if (d->gn_stats_ws && (d->hw <= 0 || d->hw % bm || d->hw / bm > UPK_GN_MAX_CHUNKS)) return 0;
  return 1;
}

extern "C" int upk_geglu_mlp_f16(upk_ctx* ctx, const upk_mlp_desc* d, upk_stream stream_) {
  if (!ctx || !d) return UPK_EINVAL;
  if (!d->x || !d->w1 || !d->w2 || !d->b1 || !d->u1 || !d->b2 || !d->residual || !d->y)
    return upk_fail(ctx, UPK_EINVAL, "geglu_mlp: null operand");
  if (!upk_geglu_mlp_supported(ctx, d))
    return upk_fail(ctx, UPK_ESHAPE, "geglu_mlp: shape outside the fused kernel's domain (c=%d inner=%d n_pad=%d)", d->c,
                    d->inner, d->n_pad);
  hipStream_t stream = (hipStream_t)stream_;
  MlpArgs s;
  memset(&s, 0, sizeof(s));
  s.x = (const f16*)d->x, s.w1 = (const f16*)d->w1, s.w2 = (const f16*)d->w2, s.zero = (const f16*)ctx->zero_page;
  s.b1 = d->b1, s.u1 = d->u1, s.b2 = d->b2, s.res = (const f16*)d->residual, s.y = (f16*)d->y;
  s.gn_cp = d->gn_stats_ws;
  s.ldx = d->ldx, s.ldr = d->ld_res, s.ldy = d->ldy, s.M = d->m;
  s.nch1 = d->c / 32, s.n1 = 2 * d->inner, s.n2 = d->n_out, s.n2pad = d->n_pad, s.nh = d->inner / 32;
  const int bm = d->rows_per_wg > 0 ? d->rows_per_wg : 64;
  s.gn_hw = d->hw > 0 ? d->hw : 1;
  s.gn_nblk = d->hw > 0 ? d->hw / bm : 1;
  s.ln_inv_dim = 1.0f / (float)(d->ln_dim > 0 ? d->ln_dim : d->c);
  s.ln_eps = d->ln_eps;
#ifdef UPK_TIMELINE
  s.dbg = getenv("UPK_MLP_TL") ? (unsigned long long*)((char*)ctx->ws + ctx->ws_bytes - 4096) : nullptr;
#endif
  const size_t lds = (size_t)(s.nch1 + s.nh) * bm * 64 + (size_t)bm * 8 + (size_t)s.n1 * 8;
  void (*fn)(const MlpArgs) = bm == 32 ? mlp_kernel<2, 7> : mlp_kernel<4, 7>;
  static unsigned long long attr32 = 0, attr64 = 0;
  if (int rc = upk_lds_attr_once(ctx, (const void*)fn, bm == 32 ? &attr32 : &attr64)) return rc;
  upk_prof_scope prof(ctx, UPK_CLS_IGEMM, stream);
  hipLaunchKernelGGL(fn, dim3((s.M + bm - 1) / bm), dim3(512), lds, stream, s);
  return upk_check_launch(ctx, "geglu_mlp");
}

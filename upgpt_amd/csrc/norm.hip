// GroupNorm(+SiLU) and LayerNorm for gfx950 — HBM/L2-bound wavefront-reduction kernels,
// fp32 statistics (GroupNorm32 computes in fp32: util.py:214-216).
//
// GroupNorm on NHWC: a group's channels are contiguous per pixel but a group spans every
// pixel, so statistics are taken in two coalesced stages:
//   1. gn_stats: grid (chunks, B); each block streams [pixels of its chunk] x [all C
//      channels] with 16-byte loads, every thread owning a fixed 8-channel vector, and
//      folds its per-channel (sum, sumsq) into the 32 groups through LDS;
//      writes partial[b][chunk][group][2] (deterministic, no global atomics).
//   2. gn_apply: every block first reduces the <=64 chunk partials of its sample into
//      per-channel scale/shift tables in LDS, then streams y = act(x*scale + shift).
// The input may be a two-source channel concat (openaimodel.py:736): groups may straddle
// the seam (e.g. 896+448 channels -> 42-wide groups), which the per-channel fold handles.
// (A single-launch variant — one workgroup per (sample, group set), two passes over its own
// strided slice — was measured at 21.8 us vs 14.6 us for this pair on the 16x16 / 32x32 UNet shapes: the
// 8-byte strided lanes from 64-256 blocks lose to two fully coalesced passes.  At <= 64 pixels per sample the
// (sample, group) fits the registers of one workgroup and one launch wins: gn_onepass_kernel.)
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int GN_MAX_CHUNKS = UPK_GN_MAX_CHUNKS;
constexpr int GN_GROUPS_MAX = UPK_GN_GROUPS_MAX;
constexpr int GN_PREF = 4;  // x vectors prefetched per thread in gn_apply

struct GnArgs {
  const f16* x1;
  const f16* x2;
  int c1, c2, ld1, ld2;
  int hw, groups, cpg;
  int nchunks, pix_per_chunk;
  const float* gamma;
  const float* beta;
  float eps;
  int silu;
  f16* y;
  int ldy;
  float* ws;  // [B][nchunks][groups][2]
  // apply pass fed by a conv epilogue's per-(row block, channel) partials: ws = [B][cp_nblk][2][cp_ld]
  int cp_nblk, cp_ld;
  const float* ws2;  // channel partials of the second source (its own producer, its own blocking)
  int cp_nblk2, cp_ld2;
};

__device__ __forceinline__ f16x8 gn_load(const GnArgs& a, long pix, int v) {
  const int ch = v * 8;
  if (ch < a.c1) return *(const f16x8*)(a.x1 + pix * a.ld1 + ch);
  return *(const f16x8*)(a.x2 + pix * a.ld2 + (ch - a.c1));
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const GnArgs a) {
  // per-(row slot, channel) partial sums; folded in a FIXED order below so that the
  // statistics (and everything downstream) are bitwise reproducible run to run.
  __shared__ float sc[2][256 * 8];
  const int C = a.c1 + a.c2;
  const int vpr = C >> 3;  // 16-byte vectors per pixel; vpr <= 256 (C <= 2048)
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x;
  const int p0 = chunk * a.pix_per_chunk;
  const int p1 = min(a.hw, p0 + a.pix_per_chunk);
  // thread layout: vpr threads cover the vectors of one pixel, rpi pixels in flight.
  const int rpi = 256 / vpr;
  const int r = tid / vpr;
  const int v = tid - r * vpr;
  if (r < rpi) {
    float s[8], ss[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
    // 8 pixels in flight per thread (a one-load-per-iteration loop runs at the latency of a load: 2 TB/s on the VAE
    // decoder's 134 MB tensors); the accumulation order per thread stays p0 + r, + rpi, ... -> same sums as before
    int p = p0 + r;
    for (; p + 7 * rpi < p1; p += 8 * rpi) {
      f16x8 xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = gn_load(a, (long)b * a.hw + p + u * rpi, v);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = (float)xv[u][j];
          s[j] += f;
          ss[j] += f * f;
        }
    }
    for (; p < p1; p += rpi) {
      const f16x8 xv = gn_load(a, (long)b * a.hw + p, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)xv[j];
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
  // fold (row slot, channel) sums into groups: 4 threads per (group, sum|sumsq), each a fixed
  // quarter of the (slot, channel) pairs, combined in a fixed order -> bitwise reproducible.
  // A group may straddle 16-byte vectors and the concat seam, hence the per-channel walk.
  {
    const int q = tid >> 2, sub = tid & 3;
    float t = 0.f;
    if (q < a.groups * 2) {
      const int g = q >> 1, which = q & 1;
      const int n = rpi * a.cpg;
      for (int e = sub; e < n; e += 4) {
        const int rr = e / a.cpg;
        const int ch = g * a.cpg + (e - rr * a.cpg);
        t += sc[which][rr * C + ch];
      }
    }
    const float t1 = t + __shfl_xor(t, 1);
    const float t2 = t1 + __shfl_xor(t1, 2);
    if (q < a.groups * 2 && sub == 0) a.ws[((long)(b * a.nchunks + chunk) * a.groups) * 2 + q] = t2;
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const GnArgs a, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float tab[];  // scale[C], shift[C]
  __shared__ float smean[GN_GROUPS_MAX], srstd[GN_GROUPS_MAX];
  const int C = a.c1 + a.c2;
  int bx, b;
  upk_xcd_xb(bx, b);
  const int tid = threadIdx.x;
  // issue this thread's first x vectors NOW: their latency overlaps the statistics /
  // table work below instead of forming a third dependent memory round trip
  const int vpr = C >> 3;
  const int p0 = bx * rows_per_block;
  const int p1 = min(a.hw, p0 + rows_per_block);
  const int nvec = (p1 - p0) * vpr;
  f16x8 pre[GN_PREF];
#pragma unroll
  for (int k = 0; k < GN_PREF; ++k) {
    const int i = tid + k * 256;
    if (i < nvec) {
      const int pr = i / vpr;
      pre[k] = gn_load(a, (long)b * a.hw + p0 + pr, i - pr * vpr);
    }
  }
  // gamma / beta of this thread's channels: cold parameters (HBM latency), requested up front
  constexpr int GB = 8;  // C <= 2048 -> <= 8 channels per thread
  float pg[GB], pb[GB];
#pragma unroll
  for (int k = 0; k < GB; ++k) {
    const int ch = tid + k * 256;
    pg[k] = ch < C ? a.gamma[ch] : 0.f;
    pb[k] = ch < C ? a.beta[ch] : 0.f;
  }
  if (a.cp_nblk > 0) {
    // per-(row block, channel) partials left by the producer conv's epilogue: fold the blocks per channel
    // (threads walk channels: coalesced), then the channels of each group, both in a fixed order
    float* chs = tab;  // [2][C] channel sums; overwritten by the scale / shift tables afterwards
    const float* w1 = a.ws + (long)b * a.cp_nblk * 2 * a.cp_ld;
    const float* w2 = a.ws2 ? a.ws2 + (long)b * a.cp_nblk2 * 2 * a.cp_ld2 : nullptr;
    for (int idx = tid; idx < 2 * C; idx += 256) {
      const int which = idx >= C ? 1 : 0;
      const int ch = idx - which * C;
      const bool second = ch >= a.c1;  // channel of the second source of the concat
      const int ld = second ? a.cp_ld2 : a.cp_ld;
      const int nblk = second ? a.cp_nblk2 : a.cp_nblk;
      const float* src = (second ? w2 + (ch - a.c1) : w1 + ch) + which * ld;
      float acc = 0.f;
#pragma unroll 8
      for (int k = 0; k < nblk; ++k) acc += src[(long)k * 2 * ld];
      chs[idx] = acc;
    }
    __syncthreads();
    __shared__ double gsum[GN_GROUPS_MAX * 2];
    if (tid < a.groups * 2) {
      const int g = tid >> 1, which = tid & 1;
      double acc = 0.0;
      for (int e = 0; e < a.cpg; ++e) acc += (double)chs[which * C + g * a.cpg + e];
      gsum[tid] = acc;
    }
    __syncthreads();
    if (tid < a.groups) {
      const double n = (double)a.hw * a.cpg;
      const double mean = gsum[tid * 2] / n;
      double var = gsum[tid * 2 + 1] / n - mean * mean;
      if (var < 0.0) var = 0.0;
      smean[tid] = (float)mean;
      srstd[tid] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
  } else {
  // fixed-order (bitwise reproducible) reduction of the chunk partials, 4 threads per
  // (group, sum|sumsq) so that the <= 64 loads per quantity are issued in parallel.
  {
    __shared__ double part[GN_GROUPS_MAX * 2 * 4];
    const int q = tid >> 2, sub = tid & 3;  // q = group*2 + which
    if (q < a.groups * 2) {
      const float* w = a.ws + (long)b * a.nchunks * a.groups * 2 + q;
      // all (<= 8) partial loads are issued before the first add: a runtime-trip-count "load, accumulate" loop is
      // compiled into that many DEPENDENT L2 round trips — and so is a conditional load (`idx < n ? w[idx] : 0` became
      // eight branches, each with its own load + s_waitcnt vmcnt(0)): the loads are unconditional on a clamped index,
      // the select follows
      float v[GN_MAX_CHUNKS / 4];
#pragma unroll
      for (int k = 0; k < GN_MAX_CHUNKS / 4; ++k) {
        const int idx = sub + 4 * k;
        v[k] = w[(long)(idx < a.nchunks ? idx : 0) * a.groups * 2];
      }
#pragma unroll
      for (int k = 0; k < GN_MAX_CHUNKS / 4; ++k) v[k] = sub + 4 * k < a.nchunks ? v[k] : 0.f;
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < GN_MAX_CHUNKS / 4; ++k) acc += (double)v[k];
      part[tid] = acc;
    }
    __syncthreads();
    if (tid < a.groups) {
      const double s = ((part[tid * 8 + 0] + part[tid * 8 + 1]) + part[tid * 8 + 2]) + part[tid * 8 + 3];
      const double ss = ((part[tid * 8 + 4] + part[tid * 8 + 5]) + part[tid * 8 + 6]) + part[tid * 8 + 7];
      const double n = (double)a.hw * a.cpg;
      const double mean = s / n;
      double var = ss / n - mean * mean;
      if (var < 0.0) var = 0.0;
      smean[tid] = (float)mean;
      srstd[tid] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
  }
  }
  __syncthreads();
  float* scale = tab;
  float* shift = tab + C;
#pragma unroll
  for (int k = 0; k < GB; ++k) {
    const int ch = tid + k * 256;
    if (ch < C) {
      const int g = ch / a.cpg;
      const float sc = srstd[g] * pg[k];
      scale[ch] = sc;
      shift[ch] = pb[k] - smean[g] * sc;
    }
  }
  __syncthreads();
  auto emit = [&](int i, const f16x8 xv) {
    const int pr = i / vpr;
    const int v = i - pr * vpr;
    const long pix = (long)b * a.hw + p0 + pr;
    const f32x4 sc0 = *(const f32x4*)(scale + v * 8), sc1 = *(const f32x4*)(scale + v * 8 + 4);
    const f32x4 sh0 = *(const f32x4*)(shift + v * 8), sh1 = *(const f32x4*)(shift + v * 8 + 4);
    f16x8 yv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = (float)xv[j] * (j < 4 ? sc0[j] : sc1[j - 4]) + (j < 4 ? sh0[j] : sh1[j - 4]);
      if (a.silu) f = upk_silu(f);
      yv[j] = (f16)f;
    }
    *(f16x8*)(a.y + pix * a.ldy + v * 8) = yv;
  };
#pragma unroll
  for (int k = 0; k < GN_PREF; ++k) {  // static register indices (no scratch)
    const int i = tid + k * 256;
    if (i < nvec) emit(i, pre[k]);
  }
  for (int base = GN_PREF * 256; base < nvec; base += GN_PREF * 256) {  // (large tensors: GN_PREF loads in flight)
#pragma unroll
    for (int k = 0; k < GN_PREF; ++k) {
      const int i = base + tid + k * 256;
      if (i < nvec) {
        const int pr = i / vpr;
        pre[k] = gn_load(a, (long)b * a.hw + p0 + pr, i - pr * vpr);
      }
    }
#pragma unroll
    for (int k = 0; k < GN_PREF; ++k) {
      const int i = base + tid + k * 256;
      if (i < nvec) emit(i, pre[k]);
    }
  }
}

// GroupNorm (+ SiLU) of a SMALL feature map in one launch: grid (groups, batch), a workgroup owns one (sample, group) and
// keeps its hw * cpg values in registers between the statistics and the normalisation — the 4x4 and 8x8 levels of the
// UNet, where the concatenated decoder inputs ([h | skip], openaimodel.py:736) have no producer that could leave the
// statistics as a by-product and the stats + apply pair cost two ~5 us launches for 0.2-0.9 MB.  A thread owns up to NV
// vectors of V consecutive channels (a vector never straddles the concat seam: c1 % V == 0); per-thread fp32 sums over
// its <= NV * V values, then fp64 across the workgroup in a fixed order (bitwise reproducible), mean / variance as in
// gn_apply_kernel.  (At 16x16 and above one workgroup per (sample, group) reads 8-byte strided lanes and loses to the
// two coalesced passes: see the header.)
template <int V, int NV>
__global__ __launch_bounds__(256) void gn_onepass_kernel(const GnArgs a) {
  typedef f16 hv __attribute__((ext_vector_type(V)));
  __shared__ double red[8];
  int g, b;
  upk_xcd_xb(g, b);
  const int tid = threadIdx.x;
  const int vpr = a.cpg / V;
  const int nvec = a.hw * vpr;
  const int c0 = g * a.cpg;
  hv x[NV];
  float gam[NV][V], bet[NV][V];
  long yoff[NV];
  bool on[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {  // every load on a clamped index, unconditional (a conditional load is a branch + wait)
    const int i = tid + k * 256;
    on[k] = i < nvec;
    const int ic = on[k] ? i : 0;
    const int p = ic / vpr;
    const int c = c0 + (ic - p * vpr) * V;
    const long pix = (long)b * a.hw + p;
    const f16* src = c < a.c1 ? a.x1 + pix * a.ld1 + c : a.x2 + pix * a.ld2 + (c - a.c1);
    x[k] = *(const hv*)src;
    yoff[k] = pix * a.ldy + c;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      gam[k][j] = a.gamma[c + j];
      bet[k][j] = a.beta[c + j];
    }
  }
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float f = on[k] ? (float)x[k][j] : 0.f;
      s += f;
      ss += f * f;
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
  const double n = (double)a.hw * a.cpg;
  const double mean = (((red[0] + red[2]) + red[4]) + red[6]) / n;
  double var = (((red[1] + red[3]) + red[5]) + red[7]) / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
  const float fmean = (float)mean;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    hv o;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float sc = rstd * gam[k][j];
      float f = (float)x[k][j] * sc + (bet[k][j] - fmean * sc);
      if (a.silu) f = upk_silu(f);
      o[j] = (f16)f;
    }
    if (on[k]) *(hv*)(a.y + yoff[k]) = o;
  }
}

// Channel partials of a producer conv with many M tiles per sample -> the group partial layout of gn_stats_kernel
// (chunk 0 holds the totals, the other chunks zero): one workgroup per (sample, group), fixed summation order.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* part, int nblk, int ld, int cpg, int groups,
                                                          int nchunks, float* ws) {
  __shared__ double red[2][4];
  const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
  const float* p = part + (long)b * nblk * 2 * ld + g * cpg;
  const int n = nblk * cpg;  // (block, channel) pairs of this group
  double s1 = 0.0, s2 = 0.0;
  for (int e0 = tid; e0 < n; e0 += 256 * 4) {
    float v1[4], v2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * 256;
      const bool on = e < n;
      const int ec = on ? e : 0;  // (unconditional loads on a clamped index: conditional ones serialise, see gn_apply_kernel)
      const int k = ec / cpg, c = ec - k * cpg;
      const float t1 = p[(long)k * 2 * ld + c], t2 = p[(long)k * 2 * ld + ld + c];
      v1[u] = on ? t1 : 0.f;
      v2[u] = on ? t2 : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s1 += (double)v1[u];
      s2 += (double)v2[u];
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = s1;
    red[1][tid >> 6] = s2;
  }
  __syncthreads();
  if (tid < nchunks * 2) {
    const int chunk = tid >> 1, which = tid & 1;
    const double t = ((red[which][0] + red[which][1]) + red[which][2]) + red[which][3];
    ws[((long)(b * nchunks + chunk) * groups + g) * 2 + which] = chunk == 0 ? (float)t : 0.f;
  }
}

// One wave per row; the row lives in registers (d <= 2048), exact two-pass mean/variance.
template <int VPL>  // 16-byte vectors per lane
__global__ __launch_bounds__(256) void layernorm_kernel(const f16* x, int ldx, int rows, int d, const float* gamma,
                                                        const float* beta, float eps, f16* y, int ldy) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = d >> 3;
  const f16* xr = x + (long)row * ldx;
  f16x8 xv[VPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int v = lane + i * 64;
    if (v < nv) {
      xv[i] = *(const f16x8*)(xr + v * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)xv[i][j];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int v = lane + i * 64;
    if (v < nv) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)xv[i][j] - mean;
        q += f * f;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q / d + eps);
  f16* yr = y + (long)row * ldy;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int v = lane + i * 64;
    if (v < nv) {
      const f32x4 g0 = *(const f32x4*)(gamma + v * 8), g1 = *(const f32x4*)(gamma + v * 8 + 4);
      const f32x4 b0 = *(const f32x4*)(beta + v * 8), b1 = *(const f32x4*)(beta + v * 8 + 4);
      f16x8 yv;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = ((float)xv[i][j] - mean) * rstd;
        yv[j] = (f16)(f * (j < 4 ? g0[j] : g1[j - 4]) + (j < 4 ? b0[j] : b1[j - 4]));
      }
      *(f16x8*)(yr + v * 8) = yv;
    }
  }
}

}  // namespace

extern "C" size_t upk_groupnorm_ws_bytes(int batch, int hw) {
  int nch, ppc;
  upk_gn_chunking(hw, &nch, &ppc);
  return (size_t)batch * nch * GN_GROUPS_MAX * 2 * sizeof(float);
}

extern "C" int upk_groupnorm_chunks(int hw) {
  int nch, ppc;
  upk_gn_chunking(hw, &nch, &ppc);
  return nch;
}

static int gn_launch(upk_ctx* ctx, const void* x1, int c1, int ld1, const void* x2, int c2, int ld2, int batch, int hw,
                     int groups, const float* gamma, const float* beta, float eps, int fuse_silu, void* y, int ldy,
                     float* stats_ws, upk_stream stream_, bool with_stats, int cp_nblk = 0, int cp_ld = 0,
                     const float* stats_ws2 = nullptr, int cp_nblk2 = 0, int cp_ld2 = 0, bool with_apply = true) {
  if (!ctx) return UPK_EINVAL;
  if (!x1 || !stats_ws || (with_apply && (!gamma || !beta || !y)))
    return upk_fail(ctx, UPK_EINVAL, "groupnorm: null pointer");
  const int C = c1 + c2;
  if (c1 <= 0 || c2 < 0 || (c1 & 7) || (c2 & 7) || (c2 && !x2) || (ld1 & 7) || (c2 && (ld2 & 7)) ||
      (with_apply && (ldy & 7)))
    return upk_fail(ctx, UPK_EINVAL, "groupnorm: channels / leading dims must be multiples of 8");
  if (groups <= 0 || groups > GN_GROUPS_MAX || C % groups || C > 2048)
    return upk_fail(ctx, UPK_ESHAPE, "groupnorm: C=%d groups=%d unsupported", C, groups);
  if (batch <= 0 || hw <= 0) return upk_fail(ctx, UPK_EINVAL, "groupnorm: empty");
  hipStream_t stream = (hipStream_t)stream_;
  GnArgs a;
  a.x1 = (const f16*)x1;
  a.x2 = (const f16*)x2;
  a.c1 = c1;
  a.c2 = c2;
  a.ld1 = ld1;
  a.ld2 = ld2;
  a.hw = hw;
  a.groups = groups;
  a.cpg = C / groups;
  upk_gn_chunking(hw, &a.nchunks, &a.pix_per_chunk);
  a.gamma = gamma;
  a.beta = beta;
  a.eps = eps;
  a.silu = fuse_silu;
  a.y = (f16*)y;
  a.ldy = ldy;
  a.ws = stats_ws;
  a.cp_nblk = cp_nblk;
  a.cp_ld = cp_ld;
  a.ws2 = stats_ws2;
  a.cp_nblk2 = cp_nblk2;
  a.cp_ld2 = cp_ld2;
  if (cp_nblk > 0 && (cp_ld < c1 || cp_nblk > GN_MAX_CHUNKS ||
                      (c2 != 0 && (!stats_ws2 || cp_ld2 < c2 || cp_nblk2 <= 0 || cp_nblk2 > GN_MAX_CHUNKS))))
    return upk_fail(ctx, UPK_EINVAL, "groupnorm apply: channel partials need stats_ld >= channels and 0 < nblk <= %d for "
                    "every source", GN_MAX_CHUNKS);
  static const bool fold1 = getenv("UPK_GN_FOLD1") && atoi(getenv("UPK_GN_FOLD1"));  // dev: timing bound of a free partial fold (WRONG results)
  if (fold1 && a.cp_nblk > 1) a.cp_nblk = 1;
  if (fold1 && a.cp_nblk2 > 1) a.cp_nblk2 = 1;
  upk_prof_scope prof(ctx, UPK_CLS_GN, stream);
  int rc = UPK_OK;
  // small feature maps: statistics and normalisation in ONE launch (gn_onepass_kernel)
  static const bool onepass_off = getenv("UPK_GN_ONEPASS") && !atoi(getenv("UPK_GN_ONEPASS"));  // dev: A/B
  if (with_stats && with_apply && hw <= 64 && !onepass_off) {
    const int v = !(a.cpg & 7) ? 8 : (!(a.cpg & 3) ? 4 : (!(a.cpg & 1) ? 2 : 0));
    const int nv = v ? (hw * (a.cpg / v) + 255) / 256 : 0;
    const dim3 grid(groups, batch);
    bool done = true;
    if (v == 8 && nv <= 2) hipLaunchKernelGGL((gn_onepass_kernel<8, 2>), grid, dim3(256), 0, stream, a);
    else if (v == 8 && nv <= 4) hipLaunchKernelGGL((gn_onepass_kernel<8, 4>), grid, dim3(256), 0, stream, a);
    else if (v == 4 && nv <= 4) hipLaunchKernelGGL((gn_onepass_kernel<4, 4>), grid, dim3(256), 0, stream, a);
    else if (v == 4 && nv <= 8) hipLaunchKernelGGL((gn_onepass_kernel<4, 8>), grid, dim3(256), 0, stream, a);
    else if (v == 2 && nv <= 8) hipLaunchKernelGGL((gn_onepass_kernel<2, 8>), grid, dim3(256), 0, stream, a);
    else done = false;
    if (done) return upk_check_launch(ctx, "gn_onepass");
  }
  if (with_stats) {
    hipLaunchKernelGGL(gn_stats_kernel, dim3(a.nchunks, batch), dim3(256), 0, stream, a);
    rc = upk_check_launch(ctx, "gn_stats");
    if (rc || !with_apply) return rc;
  }
  // apply: ~4 KB of fp16 per block (>= 3 blocks per CU on the UNet shapes: the kernel is a chain of
  // dependent memory round trips, so it needs co-resident blocks, not long per-block loops)
  // ... up to ~4096 blocks: beyond that (VAE decoder: 64 K - 524 K rows) a block's fixed part — folding the partial
  // statistics, gamma / beta, the scale / shift table — outweighs its 4 KB of data
  static const int blk_elems = getenv("UPK_GN_BLOCK_ELEMS") ? atoi(getenv("UPK_GN_BLOCK_ELEMS")) : 2048;  // dev
  int rows = (blk_elems + C - 1) / C;
  if (rows < 1) rows = 1;
  const long rows_big = ((long)hw * batch + 4095) / 4096;
  if (rows_big > rows) rows = (int)(rows_big < 4096 ? rows_big : 4096);
  const int blocks = (hw + rows - 1) / rows;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks, batch), dim3(256), (size_t)C * 2 * sizeof(float), stream, a, rows);
  return upk_check_launch(ctx, "gn_apply");
}

extern "C" int upk_groupnorm_nhwc_f16(upk_ctx* ctx, const void* x1, int c1, int ld1, const void* x2, int c2, int ld2,
                                      int batch, int hw, int groups, const float* gamma, const float* beta,
                                      float eps, int fuse_silu, void* y, int ldy, float* stats_ws,
                                      upk_stream stream) {
  return gn_launch(ctx, x1, c1, ld1, x2, c2, ld2, batch, hw, groups, gamma, beta, eps, fuse_silu, y, ldy, stats_ws,
                   stream, true);
}

extern "C" int upk_groupnorm_stats_nhwc_f16(upk_ctx* ctx, const void* x1, int c1, int ld1, const void* x2, int c2, int ld2,
                                            int batch, int hw, int groups, float* stats_ws, upk_stream stream) {
  return gn_launch(ctx, x1, c1, ld1, x2, c2, ld2, batch, hw, groups, nullptr, nullptr, 0.f, 0, nullptr, 0, stats_ws,
                   stream, true, 0, 0, nullptr, 0, 0, false);
}

extern "C" int upk_groupnorm_apply_nhwc_f16(upk_ctx* ctx, const void* x1, int c1, int ld1, const void* x2, int c2,
                                            int ld2, int batch, int hw, int groups, const float* gamma,
                                            const float* beta, float eps, int fuse_silu, void* y, int ldy,
                                            const float* stats_ws, int stats_mode, int stats_nblk, int stats_ld,
                                            const float* stats_ws2, int stats_nblk2, int stats_ld2,
                                            upk_stream stream) {
  if (stats_mode != 1 && stats_mode != 2) return upk_fail(ctx, UPK_EINVAL, "groupnorm apply: stats_mode %d", stats_mode);
  if (stats_mode == 2 && stats_nblk <= 0) return upk_fail(ctx, UPK_EINVAL, "groupnorm apply: stats_nblk %d", stats_nblk);
  if (stats_mode == 1 && c2 != 0)
    return upk_fail(ctx, UPK_EINVAL, "groupnorm apply: per-group partials (mode 1) describe a single-source tensor");
  return gn_launch(ctx, x1, c1, ld1, x2, c2, ld2, batch, hw, groups, gamma, beta, eps, fuse_silu, y, ldy,
                   (float*)stats_ws, stream, false, stats_mode == 2 ? stats_nblk : 0, stats_ld, stats_ws2, stats_nblk2,
                   stats_ld2);
}

extern "C" int upk_layernorm_f16(upk_ctx* ctx, const void* x, int ldx, int rows, int d, const float* gamma,
                                 const float* beta, float eps, void* y, int ldy, upk_stream stream_) {
  if (!ctx) return UPK_EINVAL;
  if (!x || !gamma || !beta || !y) return upk_fail(ctx, UPK_EINVAL, "layernorm: null pointer");
  if (rows <= 0 || d <= 0 || (d & 7) || (ldx & 7) || (ldy & 7) || d > 2048)
    return upk_fail(ctx, UPK_ESHAPE, "layernorm: d=%d must be a multiple of 8 and <= 2048", d);
  hipStream_t stream = (hipStream_t)stream_;
  const int nv = d >> 3;
  dim3 grid((rows + 3) / 4), block(256);
  upk_prof_scope prof(ctx, UPK_CLS_LN, stream);
  const f16* xp = (const f16*)x;
  f16* yp = (f16*)y;
  if (nv <= 64)
    hipLaunchKernelGGL((layernorm_kernel<1>), grid, block, 0, stream, xp, ldx, rows, d, gamma, beta, eps, yp, ldy);
  else if (nv <= 128)
    hipLaunchKernelGGL((layernorm_kernel<2>), grid, block, 0, stream, xp, ldx, rows, d, gamma, beta, eps, yp, ldy);
  else
    hipLaunchKernelGGL((layernorm_kernel<4>), grid, block, 0, stream, xp, ldx, rows, d, gamma, beta, eps, yp, ldy);
  return upk_check_launch(ctx, "layernorm");
}

extern "C" int upk_groupnorm_finalize_f32(upk_ctx* ctx, const float* partials, int nblk, int ld, int batch, int hw, int c,
                                          int groups, float* ws, upk_stream stream_) {
  if (!ctx) return UPK_EINVAL;
  if (!partials || !ws) return upk_fail(ctx, UPK_EINVAL, "groupnorm finalize: null pointer");
  if (nblk <= 0 || batch <= 0 || hw <= 0 || c <= 0 || groups <= 0 || groups > GN_GROUPS_MAX || c % groups || ld < c)
    return upk_fail(ctx, UPK_EINVAL, "groupnorm finalize: bad shape");
  int nchunks, ppc;
  upk_gn_chunking(hw, &nchunks, &ppc);
  hipStream_t stream = (hipStream_t)stream_;
  upk_prof_scope prof(ctx, UPK_CLS_GN, stream);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, batch), dim3(256), 0, stream, partials, nblk, ld, c / groups, groups,
                     nchunks, ws);
  return upk_check_launch(ctx, "gn_finalize");
}

// Fused attention for gfx950: softmax(Q K^T * scale) V with online softmax; the
// [B*h, n, n] score tensor of the reference (attention.py:180-191, model.py:185-195) never
// exists.  One wave owns 16 query rows; both matmuls run on v_mfma_f32_16x16x32_f16 with
// the operands arranged so that no cross-lane data movement is needed between them:
//
//   S^T[key][q]  = sum_d K[key][d] * Q[q][d]      A operand = K rows, B operand = Q rows
//        -> lane (g = lane>>4, c = lane&15) holds S^T[16t + 4g + r][c], t = 0,1, r = 0..3
//   O^T[dc][q]   = sum_k' V^T[dc][key(k')] * P^T[key(k')][q]
//        -> the B operand wants P^T[k' = 8g .. 8g+7][c]; we DEFINE key(8g + 4t + r) =
//           16t + 4g + r, i.e. exactly the 8 probabilities the lane already holds, and read
//           the V^T operand with the same permutation (two 8-byte loads per fragment).
//
// V arrives transposed ([B, heads, d, vt_ld]) straight from the projection GEMM's epilogue
// (igemm.hip, vt_* fields), K/Q are token-major with heads side by side.  K/V fragments are
// read through L1/L2 (per (b, head) they are <= 200 KB and shared by all query tiles).
#include "common.h"

namespace {

struct AttnArgs {
  const f16* q;
  const f16* k;
  const f16* vt;
  f16* o;
  int ldq, ldk, vt_ld, ldo;
  long qbs, kbs, obs;
  int heads, nq, nkv;
  float scale_log2;  // scale * log2(e)
};

// One K/V tile of NS*16 keys for the 16 queries of this wave (NS = 4 in the main loop, 2 for
// the tail).  Scores stay unscaled; softmax uses exp2(s*c - m*c) with c = scale*log2(e) folded
// into one FMA per score.  V^T fragments are requested before the softmax arithmetic so
// their latency hides under it.
template <int D, int QREG, int NS>
__device__ __forceinline__ void attn_tile(const AttnArgs& a, const f16* qrow, bool q_ok, const f16x8* qf,
                                          const f16* kbase, const f16* vbase, int kb, int g, int c,
                                          f32x4* o, float& mrun, float& lrun) {
  constexpr int KD = D / 32;
  constexpr int DT = D / 16;
  constexpr int NC = NS / 2;  // 32-key chunks for the PV MFMAs
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  f32x4 s[NS];
  const f16* kp[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int key = kb + 16 * t + c;
    kp[t] = kbase + (long)(key < a.nkv ? key : 0) * a.ldk;
  }
#pragma unroll
  for (int kd = 0; kd < KD; ++kd) {
    const f16x8 qv = QREG ? qf[kd] : (q_ok ? *(const f16x8*)(qrow + kd * 32) : zero8);
#pragma unroll
    for (int t = 0; t < NS; ++t)
      s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const f16x8*)(kp[t] + kd * 32), qv, s[t], 0, 0, 0);
  }
  // V^T operand fragments (only when they fit in registers next to the accumulators)
  constexpr bool VPRE = (D <= 128);
  f16x4 vpre[VPRE ? DT * NC * 2 : 1];
  if (VPRE) {
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const f16* vr = vbase + (long)i * 16 * a.vt_ld + kb + 32 * j;
        vpre[(i * NC + j) * 2] = *(const f16x4*)vr;
        vpre[(i * NC + j) * 2 + 1] = *(const f16x4*)(vr + 16);
      }
  }
  // lane (g, c) holds keys kb + 16t + 4g + r of query c
  float mx = -INFINITY;
  if (kb + NS * 16 > a.nkv) {  // tail tile: mask keys beyond n_kv (wave-uniform branch)
#pragma unroll
    for (int t = 0; t < NS; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (kb + 16 * t + 4 * g + r >= a.nkv) s[t][r] = -INFINITY;
  }
#pragma unroll
  for (int t = 0; t < NS; ++t) mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float cs = a.scale_log2;
  const float mnew = fmaxf(mrun, mx);  // finite: every tile has >= 1 valid key
  const float alpha = exp2f((mrun - mnew) * cs);
  mrun = mnew;
  const float mc = -mnew * cs;
  f16x8 pf[NC];
  float ps = 0.f;
#pragma unroll
  for (int t = 0; t < NS; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = exp2f(fmaf(s[t][r], cs, mc));
      ps += p;
      pf[t >> 1][(t & 1) * 4 + r] = (f16)p;
    }
  lrun = lrun * alpha + ps;
#pragma unroll
  for (int i = 0; i < DT; ++i) o[i] *= alpha;
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      f16x4 va, vb;
      if (VPRE) {
        va = vpre[(i * NC + j) * 2];
        vb = vpre[(i * NC + j) * 2 + 1];
      } else {
        const f16* vr = vbase + (long)i * 16 * a.vt_ld + kb + 32 * j;
        va = *(const f16x4*)vr;
        vb = *(const f16x4*)(vr + 16);
      }
      const f16x8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
      o[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[j], o[i], 0, 0, 0);
    }
}

template <int D, int QREG>
__global__ __launch_bounds__(256) void attn_kernel(const AttnArgs a) {
  constexpr int KD = D / 32;  // k-steps of Q K^T
  constexpr int DT = D / 16;  // output sub-tiles
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int bh = blockIdx.y;
  const int b = bh / a.heads;
  const int h = bh - b * a.heads;
  const int q0 = (blockIdx.x * 4 + wave) * 16;
  if (q0 >= a.nq) return;
  const int qi = q0 + c;
  const bool q_ok = qi < a.nq;
  const f16* qrow = a.q + b * a.qbs + (long)(q_ok ? qi : 0) * a.ldq + h * D + g * 8;
  const f16* kbase = a.k + b * a.kbs + h * D + g * 8;
  const f16* vbase = a.vt + ((long)(b * a.heads + h) * D + c) * a.vt_ld + g * 4;
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  f16x8 qf[QREG ? KD : 1];
  if (QREG) {
#pragma unroll
    for (int kd = 0; kd < KD; ++kd) qf[kd] = q_ok ? *(const f16x8*)(qrow + kd * 32) : zero8;
  }
  f32x4 o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float mrun = -INFINITY, lrun = 0.f;

  int kb = 0;
  if (D <= 128) {  // 64-key tiles while they are full; the 32-key form handles the rest
    for (; kb + 64 <= a.nkv; kb += 64) attn_tile<D, QREG, 4>(a, qrow, q_ok, qf, kbase, vbase, kb, g, c, o, mrun, lrun);
  }
  for (; kb < a.nkv; kb += 32) attn_tile<D, QREG, 2>(a, qrow, q_ok, qf, kbase, vbase, kb, g, c, o, mrun, lrun);

  lrun += __shfl_xor(lrun, 16);
  lrun += __shfl_xor(lrun, 32);
  const float inv = 1.0f / lrun;
  if (!q_ok) return;
  f16* orow = a.o + b * a.obs + (long)qi * a.ldo + h * D + g * 4;
#pragma unroll
  for (int i = 0; i < DT; ++i) {
    f16x4 ov;
#pragma unroll
    for (int r = 0; r < 4; ++r) ov[r] = (f16)(o[i][r] * inv);
    *(f16x4*)(orow + i * 16) = ov;
  }
}

}  // namespace

extern "C" int upk_attention_f16(upk_ctx* ctx, const void* q, int ldq, long long qbs, const void* k, int ldk,
                                 long long kbs, const void* vt, int vt_ld, void* out, int ldo, long long obs,
                                 int batch, int heads, int n_q, int n_kv, int d, float scale,
                                 upk_stream stream_) {
  if (!ctx) return UPK_EINVAL;
  if (!q || !k || !vt || !out) return upk_fail(ctx, UPK_EINVAL, "attention: null pointer");
  if (batch <= 0 || heads <= 0 || n_q <= 0 || n_kv <= 0) return upk_fail(ctx, UPK_EINVAL, "attention: empty");
  if ((ldq & 7) || (ldk & 7) || (vt_ld & 3) || (ldo & 3) || vt_ld < ((n_kv + 31) & ~31))
    return upk_fail(ctx, UPK_EINVAL, "attention: ldq/ldk %% 8, vt_ld %% 4 and vt_ld >= round_up(n_kv,32) required");
  hipStream_t stream = (hipStream_t)stream_;
  AttnArgs a;
  a.q = (const f16*)q;
  a.k = (const f16*)k;
  a.vt = (const f16*)vt;
  a.o = (f16*)out;
  a.ldq = ldq;
  a.ldk = ldk;
  a.vt_ld = vt_ld;
  a.ldo = ldo;
  a.qbs = qbs;
  a.kbs = kbs;
  a.obs = obs;
  a.heads = heads;
  a.nq = n_q;
  a.nkv = n_kv;
  a.scale_log2 = scale * 1.4426950408889634f;
  dim3 grid((n_q + 63) / 64, batch * heads), block(256);
  upk_prof_scope prof(ctx, UPK_CLS_ATTN, stream);
  switch (d) {
    case 32: hipLaunchKernelGGL((attn_kernel<32, 1>), grid, block, 0, stream, a); break;
    case 64: hipLaunchKernelGGL((attn_kernel<64, 1>), grid, block, 0, stream, a); break;
    case 128: hipLaunchKernelGGL((attn_kernel<128, 1>), grid, block, 0, stream, a); break;
    case 256: hipLaunchKernelGGL((attn_kernel<256, 1>), grid, block, 0, stream, a); break;
    case 512: hipLaunchKernelGGL((attn_kernel<512, 0>), grid, block, 0, stream, a); break;
    default: return upk_fail(ctx, UPK_ESHAPE, "attention: head dim %d not in {32,64,128,256,512}", d);
  }
  return upk_check_launch(ctx, "attention");
}

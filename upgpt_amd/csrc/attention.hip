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

  for (int kb = 0; kb < a.nkv; kb += 32) {
    f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
    const int key0 = kb + c, key1 = kb + 16 + c;
    const f16* k0p = kbase + (long)(key0 < a.nkv ? key0 : 0) * a.ldk;
    const f16* k1p = kbase + (long)(key1 < a.nkv ? key1 : 0) * a.ldk;
#pragma unroll
    for (int kd = 0; kd < KD; ++kd) {
      const f16x8 qv = QREG ? qf[kd] : (q_ok ? *(const f16x8*)(qrow + kd * 32) : zero8);
      const f16x8 ka = *(const f16x8*)(k0p + kd * 32);
      const f16x8 kc = *(const f16x8*)(k1p + kd * 32);
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qv, s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc, qv, s1, 0, 0, 0);
    }
    // lane holds keys kb + 4g + r (s0) and kb + 16 + 4g + r (s1) of query c
    float sv[8];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ka = kb + 4 * g + r;
      sv[r] = (ka < a.nkv) ? s0[r] * a.scale_log2 : -INFINITY;
      sv[4 + r] = (ka + 16 < a.nkv) ? s1[r] * a.scale_log2 : -INFINITY;
      mx = fmaxf(mx, fmaxf(sv[r], sv[4 + r]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mnew = fmaxf(mrun, mx);  // finite: every 32-key tile has >= 1 valid key
    const float alpha = exp2f(mrun - mnew);
    mrun = mnew;
    f16x8 pf;
    float ps = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float p = exp2f(sv[e] - mnew);
      ps += p;
      pf[e] = (f16)p;
    }
    lrun = lrun * alpha + ps;
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i] *= alpha;
    const f16* vp = vbase + kb;
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const f16* vr = vp + (long)i * 16 * a.vt_ld;
      const f16x4 va = *(const f16x4*)(vr);
      const f16x4 vb = *(const f16x4*)(vr + 16);
      const f16x8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
      o[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[i], 0, 0, 0);
    }
  }
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
    case 512: hipLaunchKernelGGL((attn_kernel<512, 0>), grid, block, 0, stream, a); break;
    default: return upk_fail(ctx, UPK_ESHAPE, "attention: head dim %d not in {32,64,128,512}", d);
  }
  return upk_check_launch(ctx, "attention");
}

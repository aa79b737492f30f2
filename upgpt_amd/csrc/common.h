// Internal header of libupk.so (gfx950 only). Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "../../include/upk.h"

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// GroupNorm partial-statistics chunking (norm.hip; the split-K reduce pass of igemm.hip writes the same layout):
// ws[((b * nchunks + chunk) * groups + g) * 2 + {sum, sumsq}], chunk = pix_per_chunk consecutive pixels.
constexpr int UPK_GN_MAX_CHUNKS = 32;
constexpr int UPK_GN_GROUPS_MAX = 32;
static inline void upk_gn_chunking(int hw, int* nchunks, int* pix_per_chunk) {
  int n = hw < UPK_GN_MAX_CHUNKS ? hw : UPK_GN_MAX_CHUNKS;
  if (n < 1) n = 1;
  const int ppc = (hw + n - 1) / n;
  *pix_per_chunk = ppc;
  *nchunks = (hw + ppc - 1) / ppc;
}

enum { UPK_CLS_IGEMM = 0, UPK_CLS_ATTN = 1, UPK_CLS_GN = 2, UPK_CLS_LN = 3, UPK_CLS_OTHER = 4 };

struct upk_prof_rec {
  hipEvent_t e0, e1;
  int cls;
};

struct upk_ctx {
  int device;
  int num_cus;
  char err[512];
  void* ws;
  size_t ws_bytes;
  void* zero_page;  // >= 256 B of zeros in HBM (padding source for direct-to-LDS loads)
  int cfg_override;
  int splitk_override;
  void* tune_flush;  // 512 MB cache-flush buffer of the cold autotuner (allocated on first use)
  int* step_done;    // upk_step_autoadvance: arrival counter of the sampler step kernels, or nullptr
  long long n_kernels;  // kernels enqueued through this context (upk_kernel_launches)
  // profiling
  int prof_on;
  std::vector<upk_prof_rec> recs;       // recorded, not yet collected
  std::vector<upk_prof_rec> free_recs;  // reusable event pairs
  double prof_ms[UPK_NUM_CLASSES];
  long long prof_n[UPK_NUM_CLASSES];
};

static inline int upk_fail(upk_ctx* ctx, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
static inline int upk_fail(upk_ctx* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define UPK_HIP(ctx, call)                                                              \
  do {                                                                                  \
    hipError_t e__ = (call);                                                            \
    if (e__ != hipSuccess)                                                              \
      return upk_fail(ctx, UPK_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                      __FILE__, __LINE__);                                              \
  } while (0)

// Brackets a kernel launch with HIP events on its own stream when profiling.
struct upk_prof_scope {
  upk_ctx* ctx;
  hipStream_t s;
  upk_prof_rec rec;
  bool on;
  upk_prof_scope(upk_ctx* c, int cls, hipStream_t st) : ctx(c), s(st), on(false) {
    if (!c || !c->prof_on) return;
    if (!c->free_recs.empty()) {
      rec = c->free_recs.back();
      c->free_recs.pop_back();
    } else {
      if (hipEventCreate(&rec.e0) != hipSuccess) return;
      if (hipEventCreate(&rec.e1) != hipSuccess) return;
    }
    rec.cls = cls;
    on = (hipEventRecord(rec.e0, s) == hipSuccess);
  }
  ~upk_prof_scope() {
    if (!on) return;
    (void)hipEventRecord(rec.e1, s);
    ctx->recs.push_back(rec);
  }
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute of a kernel: set once per (kernel, device), under
// a lock (contexts of several GPUs / threads share the library).  `mask`: one static per kernel, bit = device ordinal.
#include <mutex>
static inline int upk_lds_attr_once(upk_ctx* ctx, const void* fn, unsigned long long* mask) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  const unsigned long long bit = 1ull << (ctx->device & 63);
  if (*mask & bit) return UPK_OK;
  UPK_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  *mask |= bit;
  return UPK_OK;
}

static inline int upk_check_launch(upk_ctx* ctx, const char* what) {
  if (ctx) ++ctx->n_kernels;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return upk_fail(ctx, UPK_EHIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return UPK_OK;
}

// Workgroup -> work-item orders that keep a SAMPLE on one XCD (workgroup w runs on XCD w % 8): with the batch a multiple
// of 8, the row-chain kernels (xblock.hip, mlp.hip), the attention kernels, the GroupNorm apply / one-launch kernels and
// the normalising reduce passes all work on sample b from XCD b % 8, so each finds what its predecessor wrote for that
// sample in the local L2 instead of on the memory side of the fabric: forward 2.862 -> 2.844 ms (same-box A/B of the two
// builds).  The conv / GEMM launches keep their own XCD map (igemm.hip: fabric bytes of weights against
// activations; forcing the sample order on them costs +11 .. +110 us).  -DUPK_NO_XCD_SAMPLE: the plain orders (A/B builds).
__device__ __forceinline__ int upk_xcd_tile(int w, int n) {  // tile of workgroup w out of n row tiles in sample order
#ifndef UPK_NO_XCD_SAMPLE
  return (n & 7) ? w : (w & 7) * (n >> 3) + (w >> 3);
#else
  (void)n;
  return w;
#endif
}
// grid (nx, batch): (x, sample) of this workgroup
__device__ __forceinline__ void upk_xcd_xb(int& x, int& b) {
#ifndef UPK_NO_XCD_SAMPLE
  const int L = blockIdx.x + gridDim.x * blockIdx.y;
  b = L % (int)gridDim.y;
  x = L / (int)gridDim.y;
#else
  x = blockIdx.x;
  b = blockIdx.y;
#endif
}
// (v_rcp_f32, 1 ulp, instead of an IEEE division: the ~10-instruction div sequence per element was most of the VALU
// work of the GroupNorm apply passes; results are rounded to fp16 right after)
__device__ __forceinline__ float upk_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// exact (erf) GELU, attention.py:44 (F.gelu default)
// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, far below the fp16 output step): branch-free,
// ~14 instructions, where the device library's erff is a multi-branch piecewise evaluation — the GEGLU
// epilogue calls this 4x per fragment and is the longest epilogue on the path.
__device__ __forceinline__ float upk_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));  // (1 ulp; __frcp_rn is a full division sequence)
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(e, x);
}
__device__ __forceinline__ float upk_gelu(float v) { return 0.5f * v * (1.0f + upk_erf(v * 0.70710678118654752440f)); }
// v * gelu(g) for the GEGLU epilogues (attention.py:42-44), the same Abramowitz-Stegun erf with the algebra folded:
//   1 - erf(|x|) = q = (p(t) t) exp(-x^2),  x = g / sqrt 2,  t = 1 / (1 + 0.3275911 |x|)
//   gelu(g) = g Phi(g) = max(g, 0) - |g| q / 2        (Phi = 1 - q / 2 for g >= 0, q / 2 for g < 0)
// 13 VALU + 2 transcendental instructions per output instead of 19 + 2: the GEGLU epilogue is VALU-bound (7.3 M
// outputs per launch at the 32x32 level: ~4 us per SIMD pair).
__device__ __forceinline__ float upk_geglu_mul(float v, float g) {
  const float ag = fabsf(g);
  const float t = __builtin_amdgcn_rcpf(fmaf(ag, 0.3275911f * 0.70710678118654752440f, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(g * g * (-0.5f * 1.44269504088896340736f));
  const float q = p * t * e;
  return v * fmaf(ag * q, -0.5f, fmaxf(g, 0.0f));
}

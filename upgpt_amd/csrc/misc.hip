// Context, weight packing, layout conversion, timestep embedding, DDIM update,
// HIP-graph helpers and event profiling of libupk.so (gfx950).
#include <stdarg.h>

#include "common.h"

// ------------------------------------------------------------------ context
extern "C" int upk_version(void) { return UPK_VERSION; }

extern "C" int upk_create(upk_ctx** out, int device) {
  if (!out) return UPK_EINVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return UPK_ENODEV;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return UPK_EHIP;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return UPK_ENODEV;  // CDNA4 only, no fallbacks
  upk_ctx* c = new upk_ctx();
  c->device = device;
  c->num_cus = prop.multiProcessorCount;
  c->err[0] = 0;
  c->ws = nullptr;
  c->ws_bytes = 0;
  c->zero_page = nullptr;
  c->tune_flush = nullptr;
  c->step_done = nullptr;
  c->n_kernels = 0;
  {
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(device);
    // (the fill goes through a private non-blocking stream: a context may be created while another host thread
    //  captures a graph — execution lanes — and a legacy-stream hipMemset would try to join that capture)
    hipStream_t s0 = nullptr;
    bool ok = hipMalloc(&c->zero_page, 4096) == hipSuccess && hipStreamCreateWithFlags(&s0, hipStreamNonBlocking) == hipSuccess &&
              hipMemsetAsync(c->zero_page, 0, 4096, s0) == hipSuccess && hipStreamSynchronize(s0) == hipSuccess;
    if (s0) (void)hipStreamDestroy(s0);
    (void)hipSetDevice(cur);
    if (!ok) {
      (void)hipGetLastError();
      if (c->zero_page) (void)hipFree(c->zero_page);
      delete c;
      return UPK_EHIP;
    }
  }
  c->cfg_override = -1;
  c->splitk_override = 0;
  c->prof_on = 0;
  for (int i = 0; i < UPK_NUM_CLASSES; ++i) {
    c->prof_ms[i] = 0;
    c->prof_n[i] = 0;
  }
  *out = c;
  return UPK_OK;
}

extern "C" int upk_destroy(upk_ctx* ctx) {
  if (!ctx) return UPK_EINVAL;
  for (auto& r : ctx->recs) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  for (auto& r : ctx->free_recs) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  if (ctx->zero_page) (void)hipFree(ctx->zero_page);
  if (ctx->tune_flush) (void)hipFree(ctx->tune_flush);
  delete ctx;
  return UPK_OK;
}

extern "C" const char* upk_last_error(upk_ctx* ctx) { return ctx ? ctx->err : "null context"; }

extern "C" int upk_set_workspace(upk_ctx* ctx, void* dptr, size_t bytes) {
  if (!ctx) return UPK_EINVAL;
  if (((uintptr_t)dptr) & 15) return upk_fail(ctx, UPK_EINVAL, "workspace must be 16-byte aligned");
  ctx->ws = dptr;
  ctx->ws_bytes = dptr ? bytes : 0;
  return UPK_OK;
}

extern "C" int upk_num_cus(upk_ctx* ctx) { return ctx ? ctx->num_cus : 0; }

// ------------------------------------------------------------------ weight packing
namespace {

__global__ void pack_weight_kernel(const float* w, int cout, int cin, int kh, int kw, const int* row_map,
                                   int n_rows, const int* col_map, int cin_p, int n_pad, f16* out, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int j = (int)(idx & 31);
  const long t = idx >> 5;
  const int nrow = (int)(t % n_pad);
  const int kc = (int)(t / n_pad);
  const int k = kc * 32 + j;
  const int tap = k / cin_p;
  const int cip = k - tap * cin_p;
  const int ky = tap / kw, kx = tap - ky * kw;
  int ci = col_map ? col_map[cip] : (cip < cin ? cip : -1);
  int o = -1;
  if (nrow < n_rows) o = row_map ? row_map[nrow] : (nrow < cout ? nrow : -1);
  float v = 0.f;
  if (o >= 0 && o < cout && ci >= 0 && ci < cin) v = w[(((long)o * cin + ci) * kh + ky) * kw + kx];
  out[idx] = (f16)v;
}

}  // namespace

extern "C" size_t upk_packed_weight_bytes(int n_rows_packed, int cin_packed, int kh, int kw) {
  const size_t n_pad = ((size_t)n_rows_packed + 15) & ~(size_t)15;
  return n_pad * (size_t)cin_packed * kh * kw * sizeof(f16);
}

extern "C" int upk_pack_weight_f16(upk_ctx* ctx, const float* w, int cout, int cin, int kh, int kw,
                                   const int32_t* row_map, int n_rows_packed, const int32_t* col_map,
                                   int cin_packed, void* w_packed, upk_stream stream_) {
  if (!ctx) return UPK_EINVAL;
  if (!w || !w_packed) return upk_fail(ctx, UPK_EINVAL, "pack: null pointer");
  if (cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 || n_rows_packed <= 0 || cin_packed <= 0 || (cin_packed & 31))
    return upk_fail(ctx, UPK_EINVAL, "pack: cin_packed must be a positive multiple of 32");
  if (!col_map && cin_packed < cin) return upk_fail(ctx, UPK_EINVAL, "pack: cin_packed < cin without col_map");
  const int n_pad = (n_rows_packed + 15) & ~15;
  const long total = (long)n_pad * cin_packed * kh * kw;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, cout, cin, kh,
                     kw, row_map, n_rows_packed, col_map, cin_packed, n_pad, (f16*)w_packed, total);
  return upk_check_launch(ctx, "pack_weight");
}

// ------------------------------------------------------------------ small ops
namespace {

// util.py:151-171: freqs = exp(-ln(max_period) * j / half) in fp32, args = t * freqs,
// embedding = [cos | sin] (+ a zero column when dim is odd).
__global__ void timestep_embed_kernel(const float* t, int n, int dim, float log_max_period, f16* out, int ld) {
  const int i = blockIdx.x;
  const int half = dim >> 1;
  const float tv = t[i];
  for (int j = threadIdx.x; j < ld; j += blockDim.x) {
    float v = 0.f;
    if (j < 2 * half) {
      const int jj = (j < half) ? j : j - half;
      const float f = expf(-log_max_period * (float)jj / (float)half);
      const float arg = tv * f;
      v = (j < half) ? cosf(arg) : sinf(arg);
    }
    out[(long)i * ld + j] = (f16)v;
  }
}

__global__ void nchw_to_nhwc_kernel(const float* x, int c, int hw, f16* y, int ldy, int c_off, int zero_to, float scale,
                                    long total_pix) {
  const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;  // b*hw + p
  if (pix >= total_pix) return;
  const long b = pix / hw;
  const long p = pix - b * hw;
  f16* yo = y + pix * ldy;
  for (int ch = 0; ch < c; ++ch) yo[c_off + ch] = (f16)(x[(b * c + ch) * hw + p] * scale);
  for (int ch = c_off + c; ch < zero_to; ++ch) yo[ch] = (f16)0.f;
}

__global__ void nhwc_to_nchw_kernel(const f16* x, int ldx, int c, int hw, float* y, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over b, c, p (p fastest)
  if (idx >= total) return;
  const long p = idx % hw;
  const long t = idx / hw;
  const long ch = t % c;
  const long b = t / c;
  y[idx] = (float)x[(b * hw + p) * ldx + ch];
}

__global__ void f32_to_f16_kernel(const float* x, int rows, int cols, f16* y, int ldy) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)rows * cols) return;
  const long r = idx / cols;
  const long cidx = idx - r * cols;
  y[r * ldy + cidx] = (f16)x[idx];
}

// The sampler step kernels advance the device-side step counter themselves (upk_step_autoadvance): every block has read
// *step before it arrives at `done`; the block that arrives last bumps the counter and re-arms `done`.  One launch
// less per step than a separate increment kernel; the next kernel sees the new value across the launch boundary.
__device__ __forceinline__ void step_advance(const int* step, int* done) {
  __syncthreads();
  if (done && threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd((unsigned*)done, 1u) == gridDim.x - 1) {
      *done = 0;
      *(int*)step = *step + 1;
    }
  }
}

// ddim.py:189-203 in one launch (see include/upk.h for the coefficient table).
__global__ void ddim_step_kernel(float* x, const float* eps, const float* coefs, const float* noise, const int* step,
                                 float* pred_x0, f16* xin, int ld_xin, int c, int hw, long n, int* done) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int st = step ? *step : 0;
  if (idx < n) {
    const float* cf = coefs + 4 * st;
    const float c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3];
    const float xv = x[idx], e = eps[idx];
    const float p0 = (xv - c0 * e) * c1;
    float xp = c2 * p0 + c3 * e;
    if (noise) xp += noise[(long)st * n + idx];
    x[idx] = xp;
    if (pred_x0) pred_x0[idx] = p0;
    if (xin) {
      const long p = idx % hw;
      const long t = idx / hw;
      const long ch = t % c;
      const long b = t / c;
      xin[(b * hw + p) * ld_xin + ch] = (f16)xp;
    }
  }
  if (step) step_advance(step, done);
}

// Classifier-free guidance inside the update (ddim.py:173-178): the UNet ran on [uncond ; cond] (2*batch rows),
// eps = e_u + scale * (e_c - e_u); the new latent refreshes BOTH halves of the UNet stem input.
__global__ void ddim_step_cfg_kernel(float* x, const float* eps2, const float* coefs, const float* noise, const int* step,
                                     float* pred_x0, f16* xin, int ld_xin, int c, int hw, long n, float scale, int* done) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int st = step ? *step : 0;
  if (idx < n) {
    const float* cf = coefs + 4 * st;
    const float c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3];
    const float eu = eps2[idx], ec = eps2[n + idx];
    const float xv = x[idx], e = eu + scale * (ec - eu);
    const float p0 = (xv - c0 * e) * c1;
    float xp = c2 * p0 + c3 * e;
    if (noise) xp += noise[(long)st * n + idx];
    x[idx] = xp;
    if (pred_x0) pred_x0[idx] = p0;
    if (xin) {
      const long p = idx % hw;
      const long t = idx / hw;
      const long ch = t % c;
      const long b = t / c;
      const f16 h = (f16)xp;
      xin[(b * hw + p) * ld_xin + ch] = h;
      xin[(n / c + b * hw + p) * ld_xin + ch] = h;  // row offset batch*hw: the conditional half
    }
  }
  if (step) step_advance(step, done);
}

// One model evaluation of the PLMS sampler (plms.py:177-236), k = *step counts EVALUATIONS (S + 1 for S steps):
//   k = 0: e0 = eps; pseudo improved Euler predictor: xin <- ddim(x, e0, coef[0]); x itself stays; hist <- e0
//   k = 1: eps was evaluated at (predictor, t_1): e' = (e0 + eps) / 2; x <- ddim(x, e', coef[0])
//   k >= 2: DDIM index j = k - 1: e' = Adams-Bashforth over [eps, e_{j-1}, e_{j-2}, e_{j-3}] by available history
//           (3/2,-1/2 | 23/12,-16/12,5/12 | 55/24,-59/24,37/24,-9/24); x <- ddim(x, e', coef[j]); hist <- eps
// eps may be the two halves of a classifier-free-guidance pass (eps2 = [uncond ; cond], scale), cfg_rows = 2.
__global__ void plms_step_kernel(float* x, const float* eps, const float* coefs, const int* step, float* hist,
                                 float* pred_x0, f16* xin, int ld_xin, int c, int hw, long n, int cfg_rows, float scale,
                                 int* done) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = step ? *step : 0;
  if (idx < n) {
  const int j = k == 0 ? 0 : k - 1;
  const float* cf = coefs + 4 * j;
  const float c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3];
  float e = eps[idx];
  if (cfg_rows == 2) e = e + scale * (eps[n + idx] - e);
  float ep;
  bool commit = true;
  if (k == 0) {
    ep = e;
    commit = false;
    hist[idx] = e;  // slot 0
  } else if (k == 1) {
    ep = 0.5f * (hist[idx] + e);
  } else {
    const float h1 = hist[(long)((j - 1) % 3) * n + idx];
    if (j == 1) {
      ep = 1.5f * e - 0.5f * h1;
    } else {
      const float h2 = hist[(long)((j - 2) % 3) * n + idx];
      if (j == 2) {
        ep = (23.f * e - 16.f * h1 + 5.f * h2) * (1.f / 12.f);
      } else {
        const float h3 = hist[(long)((j - 3) % 3) * n + idx];
        ep = (55.f * e - 59.f * h1 + 37.f * h2 - 9.f * h3) * (1.f / 24.f);
      }
    }
    hist[(long)(j % 3) * n + idx] = e;
  }
  const float xv = x[idx];
  const float p0 = (xv - c0 * ep) * c1;
  const float xp = c2 * p0 + c3 * ep;
  if (commit) {
    x[idx] = xp;
    if (pred_x0) pred_x0[idx] = p0;
  }
  if (xin) {
    const long p = idx % hw;
    const long t = idx / hw;
    const long ch = t % c;
    const long b = t / c;
    const f16 h = (f16)xp;
    xin[(b * hw + p) * ld_xin + ch] = h;
    if (cfg_rows == 2) xin[(n / c + b * hw + p) * ld_xin + ch] = h;
  }
  }
  if (step) step_advance(step, done);
}

// ViT patch embedding, step 1 (CLIP image tower, Conv2d(3, width, patch, stride = patch, bias = False)): the
// non-overlapping patches of an fp32 NCHW image become the rows of an fp16 matrix [N * (H/p) * (W/p), ld] with
// k = c * p * p + py * p + px (the conv weight's own [out][c][py][px] flattening), zero padded to ld.
__global__ void patchify_kernel(const float* x, int c, int h, int w, int p, f16* out, int ld, long rows) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * ld) return;
  const long row = idx / ld;
  const int k = (int)(idx - row * ld);
  const int gw = w / p, gh = h / p;
  const int n = (int)(row / (gh * gw));
  const int pr = (int)(row - (long)n * gh * gw);
  const int gy = pr / gw, gx = pr - gy * gw;
  float v = 0.f;
  if (k < c * p * p) {
    const int ch = k / (p * p);
    const int r = k - ch * p * p;
    const int py = r / p, px = r - py * p;
    v = x[(((long)n * c + ch) * h + gy * p + py) * w + gx * p + px];
  }
  out[idx] = (f16)v;
}

// ViT token assembly: out[n, 0] = class_embedding + pos[0]; out[n, 1 + j] = patch_emb[n, j] + pos[1 + j]
__global__ void vit_assemble_kernel(const f16* patch, int ldp, const float* cls, const float* pos, int npatch, int dim,
                                    f16* out, int ld, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int d = (int)(idx % dim);
  const long t = idx / dim;
  const int tok = (int)(t % (npatch + 1));
  const long n = t / (npatch + 1);
  const float v = tok == 0 ? cls[d] : (float)patch[(n * npatch + tok - 1) * ldp + d];
  out[t * ld + d] = (f16)(v + pos[(long)tok * dim + d]);
}

// CLIPTextEmbeddings: one thread per 8 channels
__global__ void embed_tokens_kernel(const int* ids, const f16* tok, const f16* pos, int rows, int seq, int dim, int vocab,
                                    f16* out, int ld) {
  const int vpr = dim >> 3;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)rows * vpr) return;
  const int r = (int)(idx / vpr);
  const int v = (int)(idx - (long)r * vpr);
  int id = ids[r];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const f16x8 a = *(const f16x8*)(tok + (long)id * dim + v * 8);
  const f16x8 b = *(const f16x8*)(pos + (long)(r % seq) * dim + v * 8);
  f16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (f16)((float)a[j] + (float)b[j]);
  *(f16x8*)(out + (long)r * ld + v * 8) = o;
}

// y[i, :] = x[idx[i], :]  (one thread per 8 channels)
__global__ void gather_rows_kernel(const f16* x, int ldx, const int* idx, int n, int n_src, int dim, f16* y, int ldy) {
  const int vpr = dim >> 3;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n * vpr) return;
  const int i = (int)(t / vpr);
  const int v = (int)(t - (long)i * vpr);
  int r = idx[i];
  r = r < 0 ? 0 : (r >= n_src ? n_src - 1 : r);
  *(f16x8*)(y + (long)i * ldy + v * 8) = *(const f16x8*)(x + (long)r * ldx + v * 8);
}

__global__ void advance_step_kernel(int* step) { *step += 1; }

}  // namespace

extern "C" int upk_timestep_embed_f16(upk_ctx* ctx, const float* t, int n, int dim, float max_period, void* out,
                                      int ld_out, upk_stream stream_) {
  if (!ctx) return UPK_EINVAL;
  if (!t || !out || n <= 0 || dim <= 1 || ld_out < dim) return upk_fail(ctx, UPK_EINVAL, "timestep_embed: bad args");
  upk_prof_scope prof(ctx, UPK_CLS_OTHER, (hipStream_t)stream_);
  hipLaunchKernelGGL(timestep_embed_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream_, t, n, dim, logf(max_period),
                     (f16*)out, ld_out);
  return upk_check_launch(ctx, "timestep_embed");
}

extern "C" int upk_nchw_f32_to_nhwc_f16(upk_ctx* ctx, const float* x, int batch, int c, int hw, void* y, int ldy,
                                        int c_off, int zero_pad_to, float scale, upk_stream stream_) {
  if (!ctx) return UPK_EINVAL;
  if (!x || !y || batch <= 0 || c <= 0 || hw <= 0 || c_off < 0 || c_off + c > ldy || zero_pad_to > ldy)
    return upk_fail(ctx, UPK_EINVAL, "nchw_to_nhwc: bad args");
  const long total = (long)batch * hw;
  upk_prof_scope prof(ctx, UPK_CLS_OTHER, (hipStream_t)stream_);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, x,
                     c, hw, (f16*)y, ldy, c_off, zero_pad_to, scale, total);
  return upk_check_launch(ctx, "nchw_to_nhwc");
}

extern "C" int upk_nhwc_f16_to_nchw_f32(upk_ctx* ctx, const void* x, int ldx, int batch, int c, int hw, float* y,
                                        upk_stream stream_) {
  if (!ctx) return UPK_EINVAL;
  if (!x || !y || batch <= 0 || c <= 0 || hw <= 0 || ldx < c) return upk_fail(ctx, UPK_EINVAL, "nhwc_to_nchw: bad args");
  const long total = (long)batch * c * hw;
  upk_prof_scope prof(ctx, UPK_CLS_OTHER, (hipStream_t)stream_);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                     (const f16*)x, ldx, c, hw, y, total);
  return upk_check_launch(ctx, "nhwc_to_nchw");
}

extern "C" int upk_f32_to_f16(upk_ctx* ctx, const float* x, int rows, int cols, void* y, int ldy, upk_stream stream_) {
  if (!ctx) return UPK_EINVAL;
  if (!x || !y || rows <= 0 || cols <= 0 || ldy < cols) return upk_fail(ctx, UPK_EINVAL, "f32_to_f16: bad args");
  const long total = (long)rows * cols;
  upk_prof_scope prof(ctx, UPK_CLS_OTHER, (hipStream_t)stream_);
  hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, x,
                     rows, cols, (f16*)y, ldy);
  return upk_check_launch(ctx, "f32_to_f16");
}

extern "C" int upk_ddim_step_f32(upk_ctx* ctx, float* x, const float* eps, const float* coefs, const float* noise,
                                 const int32_t* step, float* pred_x0, void* xin, int ld_xin, int batch, int c, int hw,
                                 upk_stream stream_) {
  if (!ctx) return UPK_EINVAL;
  if (!x || !eps || !coefs || batch <= 0 || c <= 0 || hw <= 0) return upk_fail(ctx, UPK_EINVAL, "ddim_step: bad args");
  if (xin && ld_xin < c) return upk_fail(ctx, UPK_EINVAL, "ddim_step: ld_xin < c");
  const long n = (long)batch * c * hw;
  upk_prof_scope prof(ctx, UPK_CLS_OTHER, (hipStream_t)stream_);
  hipLaunchKernelGGL(ddim_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, x, eps,
                     coefs, noise, step, pred_x0, (f16*)xin, ld_xin, c, hw, n, ctx->step_done);
  return upk_check_launch(ctx, "ddim_step");
}

extern "C" int upk_ddim_step_cfg_f32(upk_ctx* ctx, float* x, const float* eps2, const float* coefs, const float* noise,
                                     const int32_t* step, float* pred_x0, void* xin, int ld_xin, int batch, int c, int hw,
                                     float scale, upk_stream stream_) {
  if (!ctx) return UPK_EINVAL;
  if (!x || !eps2 || !coefs || batch <= 0 || c <= 0 || hw <= 0) return upk_fail(ctx, UPK_EINVAL, "ddim_step_cfg: bad args");
  if (xin && ld_xin < c) return upk_fail(ctx, UPK_EINVAL, "ddim_step_cfg: ld_xin < c");
  const long n = (long)batch * c * hw;
  upk_prof_scope prof(ctx, UPK_CLS_OTHER, (hipStream_t)stream_);
  hipLaunchKernelGGL(ddim_step_cfg_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, x, eps2,
                     coefs, noise, step, pred_x0, (f16*)xin, ld_xin, c, hw, n, scale, ctx->step_done);
  return upk_check_launch(ctx, "ddim_step_cfg");
}

extern "C" int upk_plms_step_f32(upk_ctx* ctx, float* x, const float* eps, const float* coefs, const int32_t* step,
                                 float* hist, float* pred_x0, void* xin, int ld_xin, int batch, int c, int hw,
                                 float cfg_scale, int cfg, upk_stream stream_) {
  if (!ctx) return UPK_EINVAL;
  if (!x || !eps || !coefs || !step || !hist || batch <= 0 || c <= 0 || hw <= 0)
    return upk_fail(ctx, UPK_EINVAL, "plms_step: bad args");
  if (xin && ld_xin < c) return upk_fail(ctx, UPK_EINVAL, "plms_step: ld_xin < c");
  const long n = (long)batch * c * hw;
  upk_prof_scope prof(ctx, UPK_CLS_OTHER, (hipStream_t)stream_);
  hipLaunchKernelGGL(plms_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, x, eps,
                     coefs, step, hist, pred_x0, (f16*)xin, ld_xin, c, hw, n, cfg ? 2 : 1, cfg_scale, ctx->step_done);
  return upk_check_launch(ctx, "plms_step");
}

extern "C" int upk_patchify_nchw_f32_f16(upk_ctx* ctx, const float* x, int batch, int c, int h, int w, int patch, void* out,
                                        int ld_out, upk_stream stream_) {
  if (!ctx) return UPK_EINVAL;
  if (!x || !out || batch <= 0 || c <= 0 || patch <= 0 || h % patch || w % patch || ld_out < c * patch * patch)
    return upk_fail(ctx, UPK_EINVAL, "patchify: bad args (H, W must be multiples of the patch, ld_out >= c*p*p)");
  const long rows = (long)batch * (h / patch) * (w / patch);
  const long n = rows * ld_out;
  upk_prof_scope prof(ctx, UPK_CLS_OTHER, (hipStream_t)stream_);
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, x, c, h, w,
                     patch, (f16*)out, ld_out, rows);
  return upk_check_launch(ctx, "patchify");
}

extern "C" int upk_vit_assemble_f16(upk_ctx* ctx, const void* patch_emb, int ld_patch, const float* class_emb,
                                    const float* pos_emb, int batch, int npatch, int dim, void* out, int ld_out,
                                    upk_stream stream_) {
  if (!ctx) return UPK_EINVAL;
  if (!patch_emb || !class_emb || !pos_emb || !out || batch <= 0 || npatch <= 0 || dim <= 0 || ld_patch < dim ||
      ld_out < dim)
    return upk_fail(ctx, UPK_EINVAL, "vit_assemble: bad args");
  const long total = (long)batch * (npatch + 1) * dim;
  upk_prof_scope prof(ctx, UPK_CLS_OTHER, (hipStream_t)stream_);
  hipLaunchKernelGGL(vit_assemble_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                     (const f16*)patch_emb, ld_patch, class_emb, pos_emb, npatch, dim, (f16*)out, ld_out, total);
  return upk_check_launch(ctx, "vit_assemble");
}

extern "C" int upk_embed_tokens_f16(upk_ctx* ctx, const int32_t* ids, const void* tok_emb, const void* pos_emb, int rows,
                                    int seq, int dim, int vocab, void* out, int ld_out, upk_stream stream_) {
  if (!ctx) return UPK_EINVAL;
  if (!ids || !tok_emb || !pos_emb || !out || rows <= 0 || seq <= 0 || dim <= 0 || (dim & 7) || vocab <= 0 ||
      ld_out < dim || (ld_out & 7))
    return upk_fail(ctx, UPK_EINVAL, "embed_tokens: bad args (dim and ld_out must be multiples of 8)");
  const long n = (long)rows * (dim >> 3);
  upk_prof_scope prof(ctx, UPK_CLS_OTHER, (hipStream_t)stream_);
  hipLaunchKernelGGL(embed_tokens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, ids,
                     (const f16*)tok_emb, (const f16*)pos_emb, rows, seq, dim, vocab, (f16*)out, ld_out);
  return upk_check_launch(ctx, "embed_tokens");
}

extern "C" int upk_gather_rows_f16(upk_ctx* ctx, const void* x, int ldx, const int32_t* idx, int n, int n_src, int dim,
                                   void* y, int ldy, upk_stream stream_) {
  if (!ctx) return UPK_EINVAL;
  if (!x || !idx || !y || n <= 0 || n_src <= 0 || dim <= 0 || (dim & 7) || ldx < dim || (ldx & 7) || ldy < dim ||
      (ldy & 7))
    return upk_fail(ctx, UPK_EINVAL, "gather_rows: bad args (dim, ldx and ldy must be multiples of 8)");
  const long t = (long)n * (dim >> 3);
  upk_prof_scope prof(ctx, UPK_CLS_OTHER, (hipStream_t)stream_);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                     (const f16*)x, ldx, idx, n, n_src, dim, (f16*)y, ldy);
  return upk_check_launch(ctx, "gather_rows");
}

extern "C" long long upk_kernel_launches(upk_ctx* ctx, int reset) {
  if (!ctx) return -1;
  const long long n = ctx->n_kernels;
  if (reset) ctx->n_kernels = 0;
  return n;
}

extern "C" int upk_step_autoadvance(upk_ctx* ctx, int32_t* done) {
  if (!ctx) return UPK_EINVAL;
  ctx->step_done = done;
  return UPK_OK;
}

extern "C" int upk_advance_step(upk_ctx* ctx, int32_t* step, upk_stream stream_) {
  if (!ctx || !step) return UPK_EINVAL;
  hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, step);
  return upk_check_launch(ctx, "advance_step");
}

// ------------------------------------------------------------------ CU-partitioned streams
// A stream whose kernels may only be placed on the CUs of `mask` (hipExtStreamCreateWithCUMask -> the HSA queue's CU
// mask).  upk_probe_placement records where the workgroups of a launch on `stream` actually ran, which is how the host
// (upgpt_amd/lanes.py cu_partitions) learns the mask-bit -> XCD mapping instead of assuming one.
__global__ void probe_placement_kernel(uint32_t* out, int spin) {
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[blockIdx.x * 2] = xcc;
    out[blockIdx.x * 2 + 1] = hw;
    const long long t0 = __builtin_readcyclecounter();  // hold the CU for a moment so that the grid spreads over the mask
    while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  }
}

// shader clock against the constant 100 MHz wall clock over ~spin_wall_ticks wall ticks: the frequency the CU really runs at
// while whatever else is in flight keeps the chip at its power limit
__global__ void probe_clock_kernel(unsigned long long* out, long long spin_wall_ticks) {
  if (threadIdx.x == 0) {
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    unsigned long long w1 = w0;
    while ((long long)(w1 - w0) < spin_wall_ticks) {
      __builtin_amdgcn_s_sleep(16);
      w1 = wall_clock64();
    }
    out[0] = clock64() - c0;
    out[1] = w1 - w0;
  }
}

extern "C" int upk_probe_clock(upk_ctx* ctx, unsigned long long* out_dev, long long spin_wall_ticks, upk_stream stream) {
  if (!ctx || !out_dev || spin_wall_ticks <= 0) return upk_fail(ctx, UPK_EINVAL, "upk_probe_clock: bad argument");
  hipLaunchKernelGGL(probe_clock_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_dev, spin_wall_ticks);
  return upk_check_launch(ctx, "probe_clock");
}

extern "C" int upk_stream_create_cumask(upk_ctx* ctx, const uint32_t* mask, int nwords, upk_stream* out) {
  if (!ctx || !mask || nwords <= 0 || !out) return upk_fail(ctx, UPK_EINVAL, "upk_stream_create_cumask: bad argument");
  hipStream_t s = nullptr;
  UPK_HIP(ctx, hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask));
  *out = (upk_stream)s;
  return UPK_OK;
}

extern "C" int upk_stream_destroy(upk_ctx* ctx, upk_stream stream) {
  if (!ctx || !stream) return upk_fail(ctx, UPK_EINVAL, "upk_stream_destroy: bad argument");
  UPK_HIP(ctx, hipStreamDestroy((hipStream_t)stream));
  return UPK_OK;
}

extern "C" int upk_probe_placement(upk_ctx* ctx, uint32_t* out_dev, int nblocks, int spin_cycles, upk_stream stream) {
  if (!ctx || !out_dev || nblocks <= 0) return upk_fail(ctx, UPK_EINVAL, "upk_probe_placement: bad argument");
  hipLaunchKernelGGL(probe_placement_kernel, dim3(nblocks), dim3(64), 0, (hipStream_t)stream, out_dev, spin_cycles);
  return upk_check_launch(ctx, "probe_placement");
}

// ------------------------------------------------------------------ HIP graphs
struct upk_graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
};

extern "C" int upk_graph_begin(upk_ctx* ctx, upk_stream stream) {
  if (!ctx) return UPK_EINVAL;
  if (ctx->prof_on) return upk_fail(ctx, UPK_EINVAL, "graph capture with profiling enabled");
  UPK_HIP(ctx, hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return UPK_OK;
}

extern "C" int upk_graph_end(upk_ctx* ctx, upk_stream stream, upk_graph** out) {
  if (!ctx || !out) return UPK_EINVAL;
  *out = nullptr;
  hipGraph_t g = nullptr;
  UPK_HIP(ctx, hipStreamEndCapture((hipStream_t)stream, &g));
  hipGraphExec_t e = nullptr;
  hipError_t err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  if (err != hipSuccess) {
    (void)hipGraphDestroy(g);
    return upk_fail(ctx, UPK_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(err));
  }
  upk_graph* h = new upk_graph();
  h->graph = g;
  h->exec = e;
  *out = h;
  return UPK_OK;
}

extern "C" int upk_graph_launch(upk_ctx* ctx, upk_graph* g, upk_stream stream) {
  if (!ctx || !g) return UPK_EINVAL;
  UPK_HIP(ctx, hipGraphLaunch(g->exec, (hipStream_t)stream));
  return UPK_OK;
}

extern "C" int upk_graph_destroy(upk_ctx* ctx, upk_graph* g) {
  if (!g) return UPK_EINVAL;
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  delete g;
  return UPK_OK;
}

// ------------------------------------------------------------------ profiling
extern "C" int upk_prof_enable(upk_ctx* ctx, int on) {
  if (!ctx) return UPK_EINVAL;
  ctx->prof_on = on ? 1 : 0;
  return UPK_OK;
}

extern "C" int upk_prof_collect(upk_ctx* ctx, double* ms_host, long long* launches_host) {
  if (!ctx || !ms_host || !launches_host) return UPK_EINVAL;
  for (auto& r : ctx->recs) {
    UPK_HIP(ctx, hipEventSynchronize(r.e1));
    float ms = 0.f;
    UPK_HIP(ctx, hipEventElapsedTime(&ms, r.e0, r.e1));
    ctx->prof_ms[r.cls] += ms;
    ctx->prof_n[r.cls] += 1;
    ctx->free_recs.push_back(r);
  }
  ctx->recs.clear();
  for (int i = 0; i < UPK_NUM_CLASSES; ++i) {
    ms_host[i] = ctx->prof_ms[i];
    launches_host[i] = ctx->prof_n[i];
    ctx->prof_ms[i] = 0;
    ctx->prof_n[i] = 0;
  }
  return UPK_OK;
}

/*
 * upk.h — C ABI of libupk.so, the MI355X (gfx950) kernel library under the
 * UPGPT denoising hot path (UNetModel.forward x DDIMSampler loop -> VAE decode).
 *
 * The reference (soon-yau/upgpt) has NO native boundary: below its Python
 * modules there is only torch.nn.functional.  Every entry
 * point here therefore cites the reference *Python* call site whose ATen
 * dispatch it replaces.  Bindings: upgpt_amd/_lib.py (ctypes); INTEGRATION.md
 * shows the stub a reference maintainer would add.
 *
 * Rules of the ABI:
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers
 *     unless the name says host;
 *   - every launcher enqueues on the given hipStream_t, never synchronises,
 *     never allocates (workspace is caller-provided through upk_set_workspace),
 *     is safe to capture in a HIP graph, and returns 0 or a negative UPK_E*;
 *   - no global mutable state outside upk_ctx.  A context is NOT thread-safe:
 *     one host thread per context at a time (one context per GPU process is
 *     the intended use); different contexts are independent;
 *   - the only entry points that synchronise or allocate are the offline
 *     tools: upk_conv_autotune (times launches; with UPK_TUNE_COLD it keeps a
 *     512 MB cache-flush buffer in the context until upk_destroy) and the
 *     upk_prof_* collectors.
 *
 * Activation layout: NHWC / token-major fp16 ([B, H*W, C] == [M, C] row major,
 * leading dimension given in elements).  NCHW fp32 only at the public boundary
 * (upk_nchw_f32_to_nhwc_f16, UPK_F_OUT_NCHW_F32).
 */
#ifndef UPK_H_
#define UPK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UPK_VERSION 100 /* 0.1.0 */

/* error codes */
#define UPK_OK 0
#define UPK_EINVAL (-1)     /* bad argument (null pointer, misaligned, negative size) */
#define UPK_ESHAPE (-2)     /* shape not supported by any compiled kernel variant */
#define UPK_EWORKSPACE (-3) /* split-K needs more workspace than was provided */
#define UPK_EHIP (-4)       /* a HIP runtime call failed; see upk_last_error */
#define UPK_ENODEV (-5)     /* no gfx950 device */

typedef struct upk_ctx upk_ctx;
typedef void* upk_stream; /* hipStream_t */

int upk_version(void);
/* Binds to HIP device `device` (hipSetDevice is NOT called by launchers; the
 * caller keeps the device current). */
int upk_create(upk_ctx** out, int device);
int upk_destroy(upk_ctx* ctx);
const char* upk_last_error(upk_ctx* ctx);
/* Caller-owned scratch for split-K partial sums (fp16 slabs [z][m][n_pad], accumulated in fp32 inside each slice and
 * summed in fp32 by the reduce pass; fixed slice order, so results are deterministic).  May be NULL/0. */
int upk_set_workspace(upk_ctx* ctx, void* dptr, size_t bytes);
/* Number of compute units of the bound device (256 on MI355X). */
int upk_num_cus(upk_ctx* ctx);

/* ------------------------------------------------------------------ */
/* Weight packing (done once at load).                                  */
/* ------------------------------------------------------------------ */
/* Packed layout consumed by upk_conv2d_nhwc_f16 / upk_gemm_f16:
 *   fp16 [K/32][n_pad][32], k = (ky*kw + kx) * cin_pad + ci,
 *   cin_pad = round_up(cin, 32), n_pad = round_up(n_rows, 16), zero filled.
 * `row_map` (device int32[n_rows_packed], may be NULL = identity) gives, for
 * each packed row, the source output channel or -1 for a zero row: this is how
 * head-dim padding (28->32 ...) and the GEGLU value/gate interleave are made.
 * `col_map` (device int32[cin_packed], may be NULL) likewise maps packed input
 * channels to source input channels (-1 = zero column).
 * Source: fp32 OIHW conv weight (openaimodel.py:204,230,519,685; model.py) or
 * fp32 [out,in] Linear weight (attention.py:161-168,40,60) with kh=kw=1. */
int upk_pack_weight_f16(upk_ctx* ctx, const float* w_oihw, int cout, int cin, int kh, int kw,
                        const int32_t* row_map, int n_rows_packed, const int32_t* col_map,
                        int cin_packed, void* w_packed, upk_stream stream);
/* bytes needed for the packed weight */
size_t upk_packed_weight_bytes(int n_rows_packed, int cin_packed, int kh, int kw);

/* ------------------------------------------------------------------ */
/* Implicit-GEMM convolution / Linear (MFMA).                           */
/* ------------------------------------------------------------------ */
/* flags */
#define UPK_F_SILU 0x1         /* y = silu(acc + bias ...) (openaimodel.py:509)            */
#define UPK_F_GEGLU 0x2        /* y[:, j] = v * gelu_erf(g) (attention.py:42-44); weight    */
                               /* rows packed as [32 value | 32 gate] per 64-row block       */
#define UPK_F_OUT_F32 0x4      /* y is fp32 [M, ldy]                                         */
#define UPK_F_OUT_NCHW_F32 0x8 /* y is fp32 NCHW [B, N, Ho, Wo] (public boundary)            */
#define UPK_F_UPSAMPLE2X 0x10  /* input is nearest-2x upsampled on the fly                   */
                               /* (openaimodel.py:116 + :107; model.py:53-56)                */
#define UPK_F_PAD_ASYM 0x20    /* stride-2 conv with (0,1,0,1) padding (model.py:72-76)      */
#define UPK_F_QUICKGELU 0x40   /* y = v * sigmoid(1.702 v) after bias (CLIP text MLP, hidden_act quick_gelu) */

typedef struct upk_conv_desc {
  /* input: up to two NHWC fp16 sources concatenated along C (openaimodel.py:736,
   * ddpm.py:1568).  c1, c2 multiples of 32 (pad with zero channels); x2 may be NULL. */
  const void* x1;
  const void* x2;
  int32_t c1, c2;
  int32_t ld1, ld2; /* pixel stride in elements */
  int32_t batch;
  int32_t in_h, in_w; /* STORED spatial dims of the sources (before 2x upsample) */
  int32_t ksize;      /* 1 or 3 */
  int32_t stride;     /* 1 or 2 */
  /* weights + epilogue */
  const void* w_packed; /* upk_pack_weight_f16 layout, K = ksize^2 * (c1+c2) */
  int32_t n_out;        /* valid output columns (GEGLU: packed rows = 2*n_out)     */
  int32_t n_pad;        /* packed rows (multiple of 16)                            */
  const float* bias;    /* [n_pad] fp32 in PACKED row order, or NULL               */
  const void* residual; /* fp16 [M, ld_res] added after bias/act, or NULL          */
  int32_t ld_res;
  /* per-sample broadcast add (ResBlock emb_out, openaimodel.py:273):
   * rowvec[(step * rv_step_stride) + b * rv_batch_stride + n], fp32. */
  const float* rowvec;
  int32_t rv_batch_stride;
  int32_t rv_step_stride;
  const int32_t* step; /* device scalar, NULL => 0 */
  void* y;
  int32_t ldy;
  /* optional transposed tail: packed columns >= vt_from are written to
   * vt[((b*vt_heads + h)*vt_dhead + d)*vt_ld + tok] (V^T for upk_attention_f16),
   * with m = b*vt_tokens + tok, col - vt_from = h*vt_dhead + d.  vt==NULL: off. */
  void* vt;
  int32_t vt_from, vt_heads, vt_dhead, vt_ld, vt_tokens;
  int32_t flags;
  /* tile configuration = tune_cfg - 1 and split-K factor chosen by upk_conv_autotune
   * (0 = let the built-in cost model decide). */
  int32_t tune_cfg, tune_splitk;
  /* LayerNorm folded into a Linear (BasicTransformerBlock norm1/2/3 -> to_q|k|v / to_q / GEGLU proj,
   * attention.py:203-215): with ln_colsum != NULL the rows of x1 are the UN-normalised residual stream
   * (ksize 1, c2 == 0), the packed weight must be W * gamma (column scaled), bias must be b + W @ beta,
   * ln_colsum[n] = sum_k fp16(W*gamma)[n, k] in PACKED row order (fp32 [n_pad]); the kernel takes each
   * row's mean / variance over its first ln_dim channels (fp32, from the fp16 tiles it stages anyway)
   * and finishes  y = rstd * (x @ W'^T - mean * colsum) + bias'  before the usual epilogue. */
  const float* ln_colsum;
  float ln_eps;
  int32_t ln_dim;
  /* GroupNorm statistics of the OUTPUT as a by-product of the launch, so that the following GroupNorm can run
   * upk_groupnorm_apply_nhwc_f16 only.  With gn_stats_ws != NULL and a plain fp16 NHWC epilogue
   *   mode 1: a split-K launch's reduce pass (it touches every output element anyway) writes the per-chunk
   *           per-group partials of upk_groupnorm_nhwc_f16 (n_out % 8 == 0, 16-byte aligned rows);
   *   mode 2: an unsplit launch whose M tiles lie inside one sample writes per-(M tile, channel) partials
   *           [batch][nblk][2][n_pad] from its epilogue.
   * gn_stats_ws must hold batch * 32 * 2 * max(n_pad, 32) floats.  upk_conv_gn_fused() returns the mode (0 = none:
   * the consumer has to run the full GroupNorm) and nblk for a descriptor; both depend on the tuned / overridden /
   * cost-model (tile, split-K) choice, the same decision procedure as the launch. */
  float* gn_stats_ws;
  int32_t gn_groups;
  /* Appended 1x1 K segment — the ResBlock skip projection folded into its second conv
   * (openaimodel.py:274-275: `skip_connection(x) + h` with skip_connection = conv1x1 when channels change):
   *   y = conv_ksize(x1 | x2) + conv1x1(x3 | x4) + bias ...
   * x3 (and optionally x4, concatenated along C like x1 | x2) are NHWC fp16 sources with the OUTPUT's spatial
   * dims (stride 1, no upsample); c3, c4 multiples of 32.  The packed weight is the main weight followed by the
   * 1x1 weight along K ((ksize^2 (c1+c2) + c3+c4) / 32 chunks of [n_pad][32]); bias = sum of both biases.
   * Wave-specialised tile configurations only (the classic kernels refuse).  x3 == NULL: off. */
  const void* x3;
  const void* x4;
  int32_t c3, c4;
  int32_t ld3, ld4;
  /* GroupNorm (+ SiLU) of the OUTPUT applied by the split-K reduce pass (gn_fused mode 3): when this launch splits
   * K (tuned / cost-model choice) and its epilogue is plain (bias + timestep row vector + residual -> fp16 NHWC),
   * the reduce pass runs one workgroup per (sample, group): it sums the slabs, writes y (unless gno_skip_y: nobody
   * but the GroupNorm reads it), takes the group's mean / variance from the fp16-rounded values it holds in
   * registers and writes SiLU?(GroupNorm(y)) * gamma + beta to gno_y (row stride gno_ld) — the GroupNorm launch
   * that would follow (ResBlock in_layers / out_layers, openaimodel.py:255-275; SpatialTransformer.norm,
   * attention.py:250) is not needed.  gn_groups gives the group count.  Launches that do not split K ignore these
   * fields (upk_conv_gn_fused reports what will happen). */
  const float* gno_gamma;
  const float* gno_beta;
  void* gno_y;
  float gno_eps;
  int32_t gno_silu, gno_ld, gno_skip_y;
  /* LayerNorm row statistics handed from the launch that PRODUCES a tensor to the folded-LayerNorm Linear that reads
   * it (BasicTransformerBlock: attn1/attn2 to_out -> norm2/norm3 -> to_q / GEGLU proj, proj_in -> norm1 -> q|k|v;
   * attention.py:203-215).  Producer: ln_rows_out = [8][M][2] floats; a plain-epilogue launch that does not split K
   * writes, per output row and per column slot, the sum and the sum of squares of the fp16 values it stores
   * (upk_conv_ln_rows reports the slot count, 0 = this launch cannot).  Consumer: ln_colsum / ln_eps / ln_dim as
   * for the in-kernel fold, plus ln_rows_in (the producer's buffer) and ln_rows_slots: the row statistics are read
   * instead of being taken from the operand tile, so every tile configuration can run the GEMM. */
  float* ln_rows_out;
  const float* ln_rows_in;
  int32_t ln_rows_slots;
  /* Upsample (openaimodel.py:109-119, model.py:41-56: F.interpolate(scale_factor=2, mode="nearest") -> conv3x3 p1) as
   * four 2x2 convolutions on the LOW-resolution grid, 4/9 of the multiply-adds: with UPK_F_UPSAMPLE2X, ksize 3,
   * stride 1 and w_phase != NULL the launch computes, for phase (py, px) in {0,1}^2, output pixel (2y + py, 2x + px)
   * from low-resolution rows {y + py - 1, y + py} x columns {x + px - 1, x + px}.  w_phase = the four phase
   * weights, each packed like a 2x2 conv weight [n_pad][2][2][c1 + c2] (upk_pack_weight_f16 with kh = kw = 2), phase
   * p = 2 py + px at w_phase + p * n_pad * 4 * (c1 + c2) halfs, where tap (ty, tx) of phase (py, px) is the sum of
   * the 3x3 taps (ky, kx) with (py + ky - 1) >> 1 == py + ty - 1 and likewise for x.  Plain epilogue only (bias ->
   * fp16 NHWC / fp32); w_packed stays the 3x3 weight (used when the launch cannot take the phase form). */
  const void* w_phase;
  /* Row-block capacity of gn_stats_ws for the per-(row block, channel) partials of an unsplit launch (mode 2): 0 = 32
   * (batch * 32 * 2 * max(n_pad, 32) floats); larger values (buffer: batch * cap * 2 * max(n_pad, 32) floats) let
   * launches with many M tiles per sample — the VAE decoder's 64x64 ... 256x256 feature maps — leave their partials
   * too; a consumer then folds them with upk_groupnorm_finalize_f32 before upk_groupnorm_apply_nhwc_f16 (mode 1). */
  int32_t gn_stats_cap;
  /* Weight prefetch for the NEXT weight-consuming launch of this stream (round 6).  Inside the sampler
   * loop every weight is cold when its launch starts: 0.85 GB of weights and ~2.7 GB of activations per forward pass
   * through a 32 MB L2 and a 256 MB memory-side cache between two uses (a launch on cold weights costs 9-32 % more than on
   * warm ones, profiles/r06_weight_temperature_4_lanes.txt).  Measured (profiles/r06_next_weight_prefetch.txt): one forward
   * alone 2.83 -> 2.76 ms, with four forwards in flight 1.461 -> 1.471 ms — the host arms it only for the former.  With
   * pf_next != NULL the wave-specialised kernels' MFMA waves, idle while the first ring stage is in flight, touch one
   * 16-byte piece of every 128-byte line of pf_next[0 .. pf_bytes) (direct-to-LDS loads into the dump row group: no
   * registers, nothing to wait for), each workgroup its share: the lines are in the memory-side cache when the next
   * launch — upk_conv_link_prefetch's caller passes the packed weight of the next conv / Linear — asks for them.  Other
   * kernel families ignore the fields.  Purely a performance hint: results never depend on it. */
  const void* pf_next;
  int64_t pf_bytes;
} upk_conv_desc;

/* Replaces F.conv2d (3x3 s1/s2 p1, 1x1) / F.linear call sites:
 * ResBlock (openaimodel.py:255-275), Downsample (:158-160), Upsample (:109-119),
 * SpatialTransformer.proj_in/out (attention.py:233-248), CrossAttention
 * to_q/k/v/out (attention.py:161-168), GEGLU/FeedForward (attention.py:40,60),
 * time_embed/emb_layers (openaimodel.py:506-511,220), VAE Decoder convs
 * (model.py:462-568).  A Linear is ksize=1, batch=1, in_h=M, in_w=1. */
int upk_conv2d_nhwc_f16(upk_ctx* ctx, const upk_conv_desc* d, upk_stream stream);

/* Which GroupNorm by-product upk_conv2d_nhwc_f16(d) will leave in d->gn_stats_ws (see upk_conv_desc): *mode in
 * {0, 1, 2, 3}, *nblk = row blocks per sample for mode 2; 3 = the reduce pass applies the GroupNorm itself (gno_*) and
 * leaves no statistics.  Nothing is enqueued. */
int upk_conv_gn_fused(upk_ctx* ctx, const upk_conv_desc* d, int* mode, int* nblk);
/* Folds per-(row block, channel) partials ([batch][nblk][2][ld], any nblk) into the per-(chunk, group) layout of
 * upk_groupnorm_nhwc_f16's workspace (everything in chunk 0, zeros elsewhere), so that
 * upk_groupnorm_apply_nhwc_f16(..., stats = ws, stats_mode = 1, ...) can follow: GroupNorm of a tensor whose producer
 * conv ran more than 32 M tiles per sample, without a statistics pass over the tensor.  c channels, hw pixels. */
int upk_groupnorm_finalize_f32(upk_ctx* ctx, const float* partials, int nblk, int ld, int batch, int hw, int c,
                               int groups, float* ws, upk_stream stream);
/* Number of column slots upk_conv2d_nhwc_f16(d) will fill in d->ln_rows_out (0: none — the launch splits K across workgroups or has no
 * plain epilogue).  Nothing is enqueued. */
int upk_conv_ln_rows(upk_ctx* ctx, const upk_conv_desc* d, int* slots);

/* Fused feed-forward tail of a SpatialTransformer block — BasicTransformerBlock.norm3 -> FeedForward (GEGLU, erf GELU)
 * -> + residual (attention.py:42-64, 215) followed by SpatialTransformer.proj_out -> + x_in (attention.py:259-261) —
 * as ONE launch; the 4c-wide hidden activation stays in LDS:
 *     h   = (xn W1v^T + b1v) * gelu(xn W1g^T + b1g),  xn = LayerNorm(x)          [m, inner]
 *     y   = residual + [h | x] W2^T + b2                                          [m, n_out]
 * x: un-normalised rows [m, ldx] (c channels, c % 224 == 0); w1 / b1 / u1: the GEGLU Linear packed as for
 * upk_conv2d_nhwc_f16 with UPK_F_GEGLU and ln_colsum (W * gamma in [32 value | 32 gate] row blocks, b + W beta, column
 * sums of the fp16-rounded rows), inner = 4 c; w2: packed [n_pad rows][K = inner + c] weight whose K order is [h | x]
 * (Packer.append_1x1(P F2, P) for ff.net.2 / proj_out), b2 [n_pad]; n_pad <= 256.  rows_per_wg: 32 or 64 (0 = 64).
 * gn_stats_ws != NULL: per-(row block, channel) GroupNorm partials [m / hw][hw / rows_per_wg][2][n_pad] of y, as
 * upk_conv_desc.gn_stats_ws mode 2 (hw = rows per sample, a multiple of rows_per_wg). */
typedef struct upk_mlp_desc {
  const void* x;
  int32_t ldx, m, c, inner;
  const void* w1;
  const float* b1;
  const float* u1;
  float ln_eps;
  int32_t ln_dim;
  const void* w2;
  const float* b2;
  int32_t n_out, n_pad;
  const void* residual;
  int32_t ld_res;
  void* y;
  int32_t ldy;
  float* gn_stats_ws;
  int32_t hw, rows_per_wg;
} upk_mlp_desc;
int upk_geglu_mlp_f16(upk_ctx* ctx, const upk_mlp_desc* d, upk_stream stream);
/* 1 if upk_geglu_mlp_f16 takes the shape (channel family, LDS budget, row-block geometry), else 0. */
int upk_geglu_mlp_supported(upk_ctx* ctx, const upk_mlp_desc* d);

/* The cross-attention half of a BasicTransformerBlock (attention.py:257-260 with the context K / V precomputed) as one
 * launch:  t1 = a1 W_out1^T + b_out1 + t0;  q = LayerNorm(t1) W_q^T;  a2 = softmax(q K^T scale) V per head;
 * y = a2 W_out2^T + b_out2 + t1.   a1 [m, lda]: self-attention output, heads side by side at the padded head width d;
 * w_out1 / w_out2: packed [c rows][K = heads * d] (upk_pack_weight with the head padding as column map); w_q: packed
 * [heads * d rows][K = c] with the LayerNorm affine folded in (as ln_colsum weights); vec = [b_out1 (c) | colsum_q
 * (heads * d) | bias_q (heads * d) | b_out2 (c)] fp32, zero padded to a multiple of 256 floats; k_ctx [batch * n_kv, ldk],
 * vt_ctx [batch, heads, d, vt_ld] as upk_attention_f16 takes them (n_kv <= 96 <= vt_ld).  hw = rows per sample (a
 * multiple of rows_per_wg: 16 or 32, 0 = 32).  Shapes: heads = 8, (c, d) in {(224, 32), (448, 64)}. */
typedef struct upk_xblock_desc {
  const void* a1;
  int32_t lda, m, c, heads, d;
  const void* t0;
  int32_t ld_t0;
  const void* w_out1;
  const void* w_q;
  const void* w_out2;
  const float* vec;
  float ln_eps;
  int32_t ln_dim;
  const void* k_ctx;
  int32_t ldk, n_kv;
  const void* vt_ctx;
  int32_t vt_ld;
  float scale;
  void* y;
  int32_t ldy;
  int32_t hw, rows_per_wg;
} upk_xblock_desc;
int upk_cross_block_f16(upk_ctx* ctx, const upk_xblock_desc* d, upk_stream stream);
/* 1 if upk_cross_block_f16 takes the shape, else 0. */
int upk_cross_block_supported(upk_ctx* ctx, const upk_xblock_desc* d);

/* The head of a SpatialTransformer (attention.py:330-333, 257) as one launch: t0 = x W_in^T + b_in (x = the GroupNorm
 * output, proj_in a 1x1 conv), q | k | v = LayerNorm(t0) W_qkv^T.  w_in: packed [c rows][K = c]; w_qkv: packed
 * [3 heads d rows][K = c] in q | k | v order at the padded head width, LayerNorm affine folded in; vec = [b_in (c) |
 * colsum_qkv (3 heads d) | bias_qkv (3 heads d)] fp32, zero padded to a multiple of 256 floats.  t0 [m, ld_t0];
 * qk [m, ld_qk] = q | k; vt [m / hw, heads, d, vt_ld] = v transposed (vt_ld >= hw).  hw = rows per sample, a multiple of
 * rows_per_wg (16 or 32, 0 = 32).  Shapes: heads = 8, c = 224, d = 32. */
typedef struct upk_hblock_desc {
  const void* x;
  int32_t ldx, m, c, heads, d;
  const void* w_in;
  const void* w_qkv;
  const float* vec;
  float ln_eps;
  int32_t ln_dim;
  void* t0;
  int32_t ld_t0;
  void* qk;
  int32_t ld_qk;
  void* vt;
  int32_t vt_ld;
  int32_t hw, rows_per_wg;
  /* gn_part != NULL: x is the UN-normalised input of SpatialTransformer.norm (attention.py:330, GroupNorm without
   * SiLU) and the normalisation is applied to the tile inside the kernel, bit-identical to a upk_groupnorm_apply launch:
   * gn_part = the per-(row block, channel) partial sums [m / hw][gn_nblk][2][gn_ld] the producer of x left (gn_stats_ws
   * mode 2, gn_nblk <= 32), gn_gamma / gn_beta [c], gn_groups groups. */
  const float* gn_part;
  const float* gn_gamma;
  const float* gn_beta;
  int32_t gn_nblk, gn_ld, gn_groups;
  float gn_eps;
} upk_hblock_desc;
int upk_head_block_f16(upk_ctx* ctx, const upk_hblock_desc* d, upk_stream stream);
int upk_head_block_supported(upk_ctx* ctx, const upk_hblock_desc* d);

/* Convenience wrapper: y[M,N] = act(A[M,K] @ W^T + bias) + residual. */
int upk_gemm_f16(upk_ctx* ctx, const void* a, int lda, int m, int k, const void* w_packed,
                 int n_out, int n_pad, const float* bias, const void* residual, int ld_res,
                 void* y, int ldy, int flags, upk_stream stream);

/* Times every (tile configuration, split-K) candidate for this descriptor with HIP events on
 * `stream` (synchronises; call at plan-build time, never inside a captured region) and
 * returns the fastest pair (cfg is 0-based), its time, and the time of the cost model's
 * default choice.  The outputs of the descriptor are overwritten `reps` times per candidate. */
int upk_conv_autotune(upk_ctx* ctx, const upk_conv_desc* d, upk_stream stream, int reps, int* best_cfg,
                      int* best_splitk, float* best_us, float* default_us);

/* Forces a tile configuration / split-K factor for the next launches (tuning
 * and tests). cfg < 0 and splitk <= 0 restore the heuristic. */
int upk_conv_override(upk_ctx* ctx, int cfg, int splitk);
/* Number of compiled tile configurations, and a description of one.  Three families, in this order (tuning files
 * store indices: a family is only ever appended): the implicit-GEMM kernels ("4x4x2x2k2[wN]": classic and
 * wave-specialised), the A-stationary Linears ("as2x7p8": 1x1 only; the split-K slot means output-column passes per
 * workgroup) and the big-tile convs ("bt4x8x4x2n4": launches with at least one tile per CU, plain epilogues unless K
 * is split).  A configuration that cannot run a descriptor is refused (UPK_ESHAPE), never approximated. */
int upk_conv_num_configs(void);
const char* upk_conv_config_name(int cfg);

/* ------------------------------------------------------------------ */
/* Fused attention: softmax(Q K^T * scale) V, online softmax, no n^2 tensor. */
/* Replaces attention.py:178-192 (einsum/softmax/einsum) and model.py:180-196. */
/* ------------------------------------------------------------------ */
/* q  : fp16 [B, n_q , ldq ] head h at columns [h*d, (h+1)*d)
 * k  : fp16 [B, n_kv, ldk ] same head layout, batch stride = k_batch_stride elements
 * vt : fp16 [B, heads, d, vt_ld] (V transposed, vt_ld >= round_up(n_kv,32), zero padded)
 * out: fp16 [B, n_q , ldo ]
 * d in {32, 64, 128, 256, 512} (head dims are padded to these by weight packing). */
int upk_attention_f16(upk_ctx* ctx, const void* q, int ldq, long long q_batch_stride,
                      const void* k, int ldk, long long k_batch_stride, const void* vt, int vt_ld,
                      void* out, int ldo, long long o_batch_stride, int batch, int heads, int n_q,
                      int n_kv, int d, float scale, upk_stream stream);

/* Causal self-attention (n_q == n_kv == n, key j visible to query i iff j <= i): the CLIP text transformer of the
 * conditioning stage (ldm/modules/encoders/modules.py:137-162 -> transformers CLIPTextModel). */
int upk_attention_causal_f16(upk_ctx* ctx, const void* q, int ldq, long long q_batch_stride,
                             const void* k, int ldk, long long k_batch_stride, const void* vt, int vt_ld,
                             void* out, int ldo, long long o_batch_stride, int batch, int heads, int n,
                             int d, float scale, upk_stream stream);

/* Cross-attention with the query projection inside: out = softmax(to_q(LayerNorm(x)) K^T * scale) V —
 * CrossAttention.forward (attention.py:170-196) behind BasicTransformerBlock.norm2 (attention.py:213), for the
 * attention whose K / V come from the (step-invariant) context.  x: [batch][n_q][ldx] fp16 un-normalised rows,
 * cq = 224 * n channels of which the first ln_dim are real.  wq: fp16 [heads][d][cq], W * gamma (LayerNorm affine
 * folded in), head dim zero-padded to d, the d rows of each head stored in the order
 *   row (32 kd + 16 t + m)  holds  d = 32 kd + 8 (m >> 2) + 4 t + (m & 3)      (t = 0,1; m = 0..15)
 * so that the projection's MFMA output is the score MFMA's operand fragment as it stands; wq_colsum / wq_bias:
 * [heads * d] fp32 in natural d order, column sums of the fp16-rounded wq rows and W beta (+ bias).  d in {32, 64}.
 * k / vt / out as upk_attention_f16. */
int upk_attention_qproj_f16(upk_ctx* ctx, const void* x, int ldx, long long xbs, int cq, int ln_dim, float ln_eps,
                            const void* wq, const float* wq_colsum, const float* wq_bias, const void* k, int ldk,
                            long long kbs, const void* vt, int vt_ld, void* out, int ldo, long long obs, int batch,
                            int heads, int n_q, int n_kv, int d, float scale, upk_stream stream);

/* out[r, :] = tok_emb[ids[r], :] + pos_emb[r % seq, :]  (fp16 tables [vocab, dim] / [seq, dim], fp16 out [rows, ld]):
 * CLIPTextEmbeddings.  ids outside [0, vocab) are an error the kernel cannot report: validate on the host. */
int upk_embed_tokens_f16(upk_ctx* ctx, const int32_t* ids, const void* tok_emb, const void* pos_emb, int rows,
                         int seq, int dim, int vocab, void* out, int ld_out, upk_stream stream);

/* y[i, :] = x[idx[i], :] for i < n (fp16 rows of `dim` channels, idx = device int32[n], clamped to [0, n_src)):
 * the end-of-text pooling of `clip.model.CLIP.encode_text` (x[arange, text.argmax(-1)]) behind
 * FrozenCLIPTextEmbedder (ldm/modules/encoders/modules.py:164-198); the argmax over the token ids is host logic. */
int upk_gather_rows_f16(upk_ctx* ctx, const void* x, int ldx, const int32_t* idx, int n, int n_src, int dim,
                        void* y, int ldy, upk_stream stream);

/* CLIP image tower (ldm/modules/encoders/modules.py:234-256 -> clip.model.VisionTransformer):
 * patch embedding = upk_patchify_nchw_f32_f16 (non-overlapping p x p patches of an fp32 NCHW image -> fp16 rows
 * [batch*(H/p)*(W/p), ld_out], k = c*p*p + py*p + px, zero padded) followed by a plain upk_gemm_f16 with the
 * Conv2d(3, width, p, stride p) weight flattened; upk_vit_assemble_f16 prepends the class token and adds the
 * positional embedding: out[n, 0] = class_emb + pos[0], out[n, 1 + j] = patch_emb[n, j] + pos[1 + j]. */
int upk_patchify_nchw_f32_f16(upk_ctx* ctx, const float* x, int batch, int c, int h, int w, int patch, void* out,
                              int ld_out, upk_stream stream);
int upk_vit_assemble_f16(upk_ctx* ctx, const void* patch_emb, int ld_patch, const float* class_emb,
                         const float* pos_emb, int batch, int npatch, int dim, void* out, int ld_out,
                         upk_stream stream);

/* ------------------------------------------------------------------ */
/* Normalisation (wavefront reductions, fp32 statistics).               */
/* ------------------------------------------------------------------ */
/* GroupNorm over NHWC fp16, optional fused SiLU, input may be a 2-source
 * channel concat; output is one [B*HW, c1+c2] fp16 tensor.
 * Replaces GroupNorm32 (util.py:214-216, eps 1e-5), Normalize (attention.py:76-77
 * and model.py:38-39, eps 1e-6) and the following SiLU/swish (openaimodel.py:203,227).
 * stats_ws: fp32 scratch, >= upk_groupnorm_ws_bytes(batch, hw) bytes (its contents after the call are unspecified:
 * feature maps of <= 64 pixels are normalised by ONE launch that keeps a (sample, group) in registers and never
 * touches it; larger ones by a statistics pass that fills it and an apply pass that reads it).
 * Reproducibility: every path sums in a fixed order (reruns are bit-identical).  ACROSS paths — this call, the apply pass
 * on a producer's partial statistics (upk_groupnorm_apply_nhwc_f16), a normalising split-K reduce (gno_*) — results are
 * bit-identical for feature maps of > 64 pixels only; at <= 64 pixels the one-launch kernel sums in another order and
 * agrees to fp16 rounding (so a tensor may normalise to different bits under different tuning files, never within one). */
int upk_groupnorm_nhwc_f16(upk_ctx* ctx, const void* x1, int c1, int ld1, const void* x2, int c2,
                           int ld2, int batch, int hw, int groups, const float* gamma,
                           const float* beta, float eps, int fuse_silu, void* y, int ldy,
                           float* stats_ws, upk_stream stream);
size_t upk_groupnorm_ws_bytes(int batch, int hw);
/* First half of upk_groupnorm_nhwc_f16 only: the per-(chunk, group) partial sums [batch][nchunks][groups][2]
 * (nchunks = *nchunks of upk_groupnorm_chunks(hw)). */
int upk_groupnorm_stats_nhwc_f16(upk_ctx* ctx, const void* x1, int c1, int ld1, const void* x2, int c2, int ld2,
                                 int batch, int hw, int groups, float* stats_ws, upk_stream stream);
int upk_groupnorm_chunks(int hw);
/* Second half of upk_groupnorm_nhwc_f16 only: stats_ws already holds the partial sums of x1, left by the conv
 * launch that produced it (upk_conv_desc.gn_stats_ws): stats_mode / stats_nblk as reported by upk_conv_gn_fused,
 * stats_ld = that launch's n_pad.  A two-source (concat) input needs mode 2 for BOTH sources: stats_ws2 /
 * stats_nblk2 / stats_ld2 describe x2's producer (NULL / 0 / 0 for a single source). */
int upk_groupnorm_apply_nhwc_f16(upk_ctx* ctx, const void* x1, int c1, int ld1, const void* x2, int c2,
                                 int ld2, int batch, int hw, int groups, const float* gamma,
                                 const float* beta, float eps, int fuse_silu, void* y, int ldy,
                                 const float* stats_ws, int stats_mode, int stats_nblk, int stats_ld,
                                 const float* stats_ws2, int stats_nblk2, int stats_ld2, upk_stream stream);

/* LayerNorm over the last dim of fp16 [rows, d] (attention.py:203-205, eps 1e-5). */
int upk_layernorm_f16(upk_ctx* ctx, const void* x, int ldx, int rows, int d, const float* gamma,
                      const float* beta, float eps, void* y, int ldy, upk_stream stream);

/* ------------------------------------------------------------------ */
/* Small ops of the loop.                                               */
/* ------------------------------------------------------------------ */
/* timestep_embedding (util.py:151-171): out[i] = [cos(t_i f_j) | sin(t_i f_j)],
 * f_j = exp(-ln(max_period) j / half), written as fp16 [n, ld_out] (cols >= dim zero). */
int upk_timestep_embed_f16(upk_ctx* ctx, const float* t, int n, int dim, float max_period,
                           void* out, int ld_out, upk_stream stream);

/* NCHW fp32 -> NHWC fp16 with channel offset/padding: writes channels
 * [c_off, c_off+c) of y[B, HW, ldy]; if zero_pad_to > c_off+c also zeroes the
 * channels up to zero_pad_to (UNet stem input = cat[x, person_mask], ddpm.py:1568). */
int upk_nchw_f32_to_nhwc_f16(upk_ctx* ctx, const float* x, int batch, int c, int hw, void* y,
                             int ldy, int c_off, int zero_pad_to, float scale,
                             upk_stream stream);
int upk_nhwc_f16_to_nchw_f32(upk_ctx* ctx, const void* x, int ldx, int batch, int c, int hw,
                             float* y, upk_stream stream);
/* fp32 [rows, cols] -> fp16 [rows, ldy] (context tokens). */
int upk_f32_to_f16(upk_ctx* ctx, const float* x, int rows, int cols, void* y, int ldy,
                   upk_stream stream);

/* One DDIM update (ddim.py:189-203), fp32 NCHW, n = B*C*H*W elements:
 *   coef = coefs + 4 * (*step):  {sqrt(1-a_t), 1/sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev-sigma^2)}
 *   pred_x0 = (x - c0*e) * c1 ; x_prev = c2*pred_x0 + c3*e + noise_scaled
 *   noise (may be NULL) = sigma_t * temperature * randn, [n_steps, n] indexed by *step.
 * Also refreshes the UNet stem input: xin (fp16 NHWC [B, HW, ld_xin]) channels
 * [0, C) = x_prev (the concat channels after them are static).  x is updated in place. */
int upk_ddim_step_f32(upk_ctx* ctx, float* x, const float* eps, const float* coefs,
                      const float* noise, const int32_t* step, float* pred_x0, void* xin,
                      int ld_xin, int batch, int c, int hw, upk_stream stream);
/* The same update with classifier-free guidance folded in (ddim.py:173-178): the UNet ran on 2*batch rows,
 * [unconditional ; conditional]; eps2 is [2*batch, C, HW], e = e_u + scale * (e_c - e_u); x / pred_x0 / noise are
 * [batch, ...]; x_prev refreshes BOTH halves of xin (fp16 NHWC [2*batch, HW, ld_xin]). */
int upk_ddim_step_cfg_f32(upk_ctx* ctx, float* x, const float* eps2, const float* coefs,
                          const float* noise, const int32_t* step, float* pred_x0, void* xin,
                          int ld_xin, int batch, int c, int hw, float scale, upk_stream stream);
/* One model evaluation of the PLMS sampler (ldm/models/diffusion/plms.py:177-236; eta = 0), *step = evaluation
 * counter k (S + 1 evaluations for S steps: the first step evaluates twice, pseudo improved Euler):
 *   k = 0: predictor x~ = ddim(x, e0, coef[0]) goes to xin only; k = 1: e' = (e0 + eps)/2, x <- ddim(x, e', coef[0]);
 *   k >= 2: e' = Adams-Bashforth of order min(k-1, 3)+1 over eps and the history; x <- ddim(x, e', coef[k-1]).
 * hist: fp32 [3, n] eps history ring (owned by the caller, no initialisation needed); coefs as for upk_ddim_step_f32
 * (one row per DDIM index); cfg != 0: eps holds [uncond ; cond] (2*batch) and e = e_u + cfg_scale*(e_c - e_u),
 * xin has 2*batch*hw rows. */
int upk_plms_step_f32(upk_ctx* ctx, float* x, const float* eps, const float* coefs, const int32_t* step,
                      float* hist, float* pred_x0, void* xin, int ld_xin, int batch, int c, int hw,
                      float cfg_scale, int cfg, upk_stream stream);
/* With done != NULL the sampler step kernels launched afterwards (upk_ddim_step_f32, upk_ddim_step_cfg_f32,
 * upk_plms_step_f32) add 1 to *step THEMSELVES once every workgroup has read it (done: a zero-initialised device
 * int32 they use as arrival counter and leave at zero) — one launch less per sampler step than upk_advance_step.
 * done == NULL restores the plain behaviour.  Host-side state of the context: set it around the calls. */
int upk_step_autoadvance(upk_ctx* ctx, int32_t* done);

/* Number of GPU kernels enqueued through this context since creation / the last reset (a split-K conv is two, a
 * GroupNorm with its own statistics pass is two): what bench.py reports as kernels per UNet forward. */
long long upk_kernel_launches(upk_ctx* ctx, int reset);

/* *step += 1 (end of a captured step graph). */
int upk_advance_step(upk_ctx* ctx, int32_t* step, upk_stream stream);

/* ------------------------------------------------------------------ */
/* CU-partitioned streams (execution lanes on disjoint CU sets).         */
/* The reference has no counterpart: it runs one batch on `cuda:0`       */
/* (app.py:21); lanes are this build's serving mode. */
/* ------------------------------------------------------------------ */
/* Creates a stream whose kernels are only placed on the CUs whose bit is set in `mask` (nwords 32-bit words, bit b of
 * word w = CU-mask bit 32 w + b of the HSA queue; how bits map to XCDs is measured with upk_probe_placement, not
 * assumed).  The caller owns the stream (upk_stream_destroy).  Never synchronises. */
int upk_stream_create_cumask(upk_ctx* ctx, const uint32_t* mask, int nwords, upk_stream* out);
int upk_stream_destroy(upk_ctx* ctx, upk_stream stream);
/* Launches `nblocks` one-wave workgroups on `stream`; workgroup i writes {HW_REG_XCC_ID, HW_REG_HW_ID} to
 * out_dev[2 i], out_dev[2 i + 1] (device memory, 8 nblocks bytes) and holds its CU for ~spin_cycles shader clocks so that
 * the grid spreads over every CU the stream may use.  Graph-capturable. */
int upk_probe_placement(upk_ctx* ctx, uint32_t* out_dev, int nblocks, int spin_cycles, upk_stream stream);
/* One wave spins for ~spin_wall_ticks ticks of the constant 100 MHz wall clock and writes {shader-clock cycles elapsed,
 * wall ticks elapsed} to out_dev[0..1] (two uint64): shader MHz = 100 * out[0] / out[1] — the clock the chip sustains under
 * whatever else is running (bench.py reports it for the timed configuration). */
int upk_probe_clock(upk_ctx* ctx, unsigned long long* out_dev, long long spin_wall_ticks, upk_stream stream);

/* ------------------------------------------------------------------ */
/* HIP graph helpers (the 50-step loop replays one captured step).      */
/* ------------------------------------------------------------------ */
typedef struct upk_graph upk_graph;
int upk_graph_begin(upk_ctx* ctx, upk_stream stream);
int upk_graph_end(upk_ctx* ctx, upk_stream stream, upk_graph** out);
int upk_graph_launch(upk_ctx* ctx, upk_graph* g, upk_stream stream);
int upk_graph_destroy(upk_ctx* ctx, upk_graph* g);

/* ------------------------------------------------------------------ */
/* Per-kernel-class timing with HIP events on the launch stream          */
/* (bench.py roofline).  Classes: 0 igemm, 1 attention, 2 groupnorm,     */
/* 3 layernorm, 4 other.                                                  */
/* ------------------------------------------------------------------ */
#define UPK_NUM_CLASSES 5
int upk_prof_enable(upk_ctx* ctx, int on);
/* Synchronises the recorded events and returns accumulated ms / launches
 * per class since the last reset (host arrays of UPK_NUM_CLASSES). */
int upk_prof_collect(upk_ctx* ctx, double* ms_host, long long* launches_host);

#ifdef __cplusplus
}
#endif
#endif /* UPK_H_ */

/* svdpipe.h – C ABI of libsvdpipe_hip.so, the MI355X (gfx950) kernel library under the
 * per-step SVD UNet hot path.
 *
 * The reference (inai17ibar/video-diffusion-pipeline-parallel) defines NO FFI: its hot path is the
 * Python call `latent = self.model(latent, step)` (/root/reference/src/pipeline/pipeline.py:95)
 * -> `StableVideoUNet.forward` (/root/reference/src/models/svd_unet.py:351-439)
 * -> `self.unet(sample=..., timestep=..., encoder_hidden_states=..., added_time_ids=...)`
 *    (svd_unet.py:389,400,416), where `unet` is diffusers' UNetSpatioTemporalConditionModel running
 *    on cuDNN/cuBLAS/xformers.  The entry points below are what a binding for that path binds
 *    instead of those vendor libraries; each comment names the diffusers/torch op it replaces and
 *    the reference line that reaches it.  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes; `stream` is a hipStream_t passed as void* (NULL = default stream)
 *   - device pointers are owned by the caller (PyTorch allocator); the library never allocates,
 *     frees or synchronises; every call only enqueues kernels on `stream`
 *   - return 0 on success, negative SP_E* on a rejected argument (nothing is launched);
 *     sp_last_error() returns a thread-local message
 *   - activations are fp16, channels-last: a "token matrix" [rows][C] where rows enumerate
 *     (frame, y, x) in that order (NHWC per frame); weights are fp16 [N][K] (K contiguous);
 *     biases / norm affine parameters are fp32
 *   - gfx950 code objects only
 */
#ifndef SVDPIPE_H
#define SVDPIPE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_OK 0
#define SP_EINVAL (-1)   /* bad argument / unsupported shape */
#define SP_ELAUNCH (-2)  /* hipLaunchKernel reported an error */

const char *sp_last_error(void);
int sp_version(void);
/* bytes of zero-filled device memory every gather kernel needs behind `zero_page` */
#define SP_ZERO_PAGE_BYTES 4096

/* ---------------------------------------------------------------------------------------------
 * Implicit-GEMM family:  D[M][Nout] = epilogue( sum_{tap,k} A_tap[m][k] * W[n][tap*Cin + k] )
 * Replaces torch.nn.functional.conv2d (ResnetBlock2D conv1/conv2, Downsample2D, Upsample2D conv,
 * conv_in/conv_out), conv3d with kernel (3,1,1) (TemporalResnetBlock), 1x1 shortcut convs and every
 * nn.Linear of the transformer blocks – i.e. cuDNN/cuBLAS under svd_unet.py:416.
 * MFMA fp16 -> fp32 accumulate; A tiles are gathered straight from the NHWC tensor into LDS
 * (no im2col buffer).
 * ------------------------------------------------------------------------------------------- */
enum { SP_A_LINEAR = 0, SP_A_CONV3X3 = 1, SP_A_TEMPORAL3 = 2 };

typedef struct sp_gemm_desc {
  /* A operand */
  const void *a;        /* fp16 [rows_in][lda] */
  int64_t lda;          /* elements between consecutive A rows (>= cin) */
  int mode;             /* SP_A_* */
  int cin;              /* channels per tap (multiple of 64); K = taps*cin */
  /* geometry for SP_A_CONV3X3: input images [n_img][hin][win], output [n_img][hout][wout];
     pad 1; stride 1 or 2; upsample2x!=0 reads the input through a nearest-neighbour x2 upsample */
  int n_img, hin, win, hout, wout, stride, upsample2x;
  /* geometry for SP_A_TEMPORAL3: rows = [batch][frames][hw]; taps are frames f-1, f, f+1 */
  int frames; int64_t hw;
  /* B operand */
  const void *w;        /* fp16 [n][taps*cin]; for geglu the rows are pre-interleaved in blocks of 16 */
  int m, n;             /* GEMM M (output rows) and N (weight rows; multiple of 64) */
  /* epilogue: v = oscale*(acc + bias[n] + bias2[(m/bias2_rows)][n]) + r1scale*res1 + r2scale*res2
     geglu!=0: out[m][j] = h*gelu(gate) over the interleaved column pairs, Nout = n/2
     (bias applies before gelu; oscale/res are applied to the product) */
  const float *bias;    /* [n] or NULL */
  const float *bias2;   /* [nb][ldb2] or NULL */
  int64_t bias2_rows;   /* rows of D sharing one bias2 row (e.g. frames*H*W of a batch item) */
  int64_t ldb2;         /* floats between bias2 rows (0 = n) */
  const void *res1; int64_t ldr1; float r1scale;   /* fp16 [m][ldr1] or NULL */
  const void *res2; int64_t ldr2; float r2scale;
  float oscale;
  int geglu;
  int n_store;          /* number of leading output columns actually stored (<= Nout); 0 = all */
  void *d; int64_t ldd; /* fp16 [m][ldd] */
  const void *zero_page;
  /* LayerNorm folded into this contraction (SP_A_LINEAR only, bias2 must be NULL): with W pre-multiplied by the norm's
     gamma, LN(x).W^T + b = rstd[m]*(x.W'^T - mean[m]*ln_colsum[n]) + bias[n], so the normalised tensor is never
     written: ln_stats = fp32 [m][2] (mean, rstd) from sp_ln_stats_f16, ln_colsum[n] = sum_k W'[n][k] (of the fp16
     values), bias[n] = W.beta + b.  NULL = no fold. */
  const float *ln_stats;
  const float *ln_colsum;
  /* Guidance mix + Euler update folded into the epilogue of the UNet's last convolution (conv_out: n = 64 padded
     weight rows of which 4 are real; /root/reference/src/models/svd_unet.py:410-439).  euler_out != NULL: instead of
     storing eps rows in d, row m = (b, f, pixel) updates the four latent channels
       eps = euler_eps_uncond ? u + g[f]*(eps - u) (evaluated in fp16 like the reference) : eps
       x0 = eps*(-sigma/sqrt(sigma^2+1)) + x/(sigma^2+1);  x' = x + (x - x0)/sigma*(sigma_next - sigma)   (fp32)
     with x read from euler_latent and x' written to euler_out, both fp16 (B,4,F,H,W); d is not written. */
  const void *euler_latent; void *euler_out;
  const void *euler_eps_uncond; int64_t euler_ld_eps;   /* fp16 [m][euler_ld_eps] eps rows of the unconditional pass, or NULL */
  const float *euler_guidance;                         /* fp32 [frames] per-frame guidance scale (with euler_eps_uncond) */
  float euler_sigma, euler_sigma_next;
  int euler_frames; int64_t euler_hw;                  /* m = b*frames*hw + f*hw + pixel */
  /* Optional scratch (the library never allocates): with at least sp_gemm_workspace_bytes(desc) bytes, contractions with
     few rows and a long K (m <= 6144, K >= 8192: the 3x3 convolutions of the UNet's 2,016-row level, 4,032 rows for a micro-batch of two) are split over K on 256 x 256 tiles,
     fp32 partial sums go here and a second kernel reduces them and applies the epilogue.  NULL = never split. */
  void *workspace; size_t workspace_bytes;
  /* LayerNorm statistics of the OUTPUT rows, for the next contraction's ln_stats (what a following sp_ln_stats_f16 pass
     over d would compute, without that pass): ln_out = fp32 [m][2] (mean, rstd = 1/sqrt(var + ln_out_eps)) of the n
     stored fp16 values of every row.  A row must be one to four 256- or 320-column tiles (n = 256 ... 1280; with more
     than one tile the per-tile sums pass through `workspace`, >= m * tiles * 8 bytes, 8-byte aligned, and a second small
     kernel folds them); no geglu, no
     n_store, no Euler tail; the call runs on the ping-pong kernels (no split-K).  Sums are folded in a fixed order:
     bit-reproducible.  NULL = off. */
  float *ln_out; float ln_out_eps;
  /* Per-row-group weights (SP_A_LINEAR only): with w_group_rows > 0, output rows [g*w_group_rows, (g+1)*w_group_rows)
     are multiplied with the weight matrix at w + g*w_group_stride halves instead of w (what a GroupNorm folded into the
     linear layer behind it needs: one scaled copy of the weights per frame, sp_groupnorm_fold_linear_f16; combine with a
     bias2 row per group).  w_group_rows must be a multiple of 128 (no tile may straddle two groups); the call runs on the
     ping-pong kernels (no split-K).  0 = one weight matrix for every row. */
  int64_t w_group_rows; int64_t w_group_stride;
  /* GroupNorm statistics of the NEXT norm out of this contraction's epilogue (round 5): gn_part = fp32
     [m/256][2][n][2] -- for every 256-row tile, each of its two 128-row halves and every output column, (sum, sum of
     squares) of the fp32 output values of the half's rows (bias / bias2 included, before the rounding to fp16).
     With residuals the sums are of the FINAL stored fp16 values instead (the tile is rebuilt in LDS behind the stores).
     sp_groupnorm_tile_sums_f16 folds them into the (mean, rstd) of any instance that is a whole number of tiles and
     normalises d without a statistics pass over it.  Needs m a multiple of 256, n a multiple of 256 or 320, no geglu /
     folded LayerNorm / ln_out / n_store / Euler tail; the call runs on the 256-row ping-pong tiles.  Sums
     are folded in a fixed order (bit-reproducible).  NULL = off. */
  float *gn_part;
  /* Extra LINEAR tap (round 5): behind the taps of `mode` the contraction runs on over cin2 channels of a SECOND tensor a2
     (fp16 [m][lda2], row i for output row i), multiplied with weight columns [taps*cin, taps*cin + cin2): w is then
     [n][taps*cin + cin2].  A resnet's 1x1 shortcut convolution folded into its second 3x3 convolution
     (conv2(h) + conv_shortcut(x) = one contraction with bias = b2 + b_sc): the skip tensor is neither written nor read.
     cin2 a multiple of 64; runs on the 256-row ping-pong tiles (n a multiple of 256 or 320; no geglu / folded LayerNorm /
     ln_out / n_store / Euler tail / per-group weights / split-K).  NULL = off. */
  const void *a2; int64_t lda2; int cin2;
} sp_gemm_desc;

int sp_gemm_f16(const sp_gemm_desc *desc, void *stream);
/* sizeof(sp_gemm_desc) as the library was built: a binding in another language checks its mirror of the struct against
 * it once at load time (the Python one does: hip/__init__.py::load). */
size_t sp_gemm_desc_size(void);
/* bytes of sp_gemm_desc.workspace this contraction can use (0: it is not a split-K candidate) */
size_t sp_gemm_workspace_bytes(const sp_gemm_desc *desc);

/* Name of the kernel instantiation the calling thread's last sp_gemm_f16 launched (e.g. "gemm_pp_kernel<256, 320, 0>";
 * thread-local, valid until that thread's next call): lets a profile attribute FLOPs to kernel templates. */
const char *sp_gemm_last_kernel(void);

/* Test / micro-benchmark hook (no counterpart in the reference): pins the kernel family sp_gemm_f16 picks for the
 * shapes that family supports; everything else keeps the automatic choice.  Process-wide, not thread-safe: set it
 * before the calls it should affect.  route 0 = automatic (default), 1 = small tiles only, 2 = ping-pong large tiles
 * (bm in {0,128,192,256}, bn in {0,256,320}; 0 = automatic), 3 = persistent-stream tiles (bm in {0,192,256},
 * bn in {0,256}), 4 = split-K whenever a workspace is given (also for short K). */
int sp_gemm_set_route(int route, int bm, int bn);

/* y[n] = act_out( W[n][:] . act_in(x) + b[n] ), M = 1.  Replaces the nn.Linear GEMVs of the
 * timestep / added-time / frame-position embeddings and every `time_emb_proj` (diffusers
 * TimestepEmbedding, ResnetBlock2D.time_emb_proj).  x, W fp16; b, y fp32 (y_f16 optional copy).
 * silu_in / silu_out apply SiLU to the input vector / output. `rows` independent input vectors. */
int sp_gemv_f16(const void *x, int64_t ldx, const void *w, const float *b, float *y, void *y_f16,
                int64_t ldy, int rows, int n, int k, int silu_in, int silu_out, void *stream);
/* `batch` independent GEMVs of one shape in a single launch (the 32 single-token cross-attention modules and the 16
 * frame-position MLPs of a forward are grouped by width).  Strides are in elements between consecutive problems;
 * x_stride = 0 shares the input, b may be NULL.  Otherwise as sp_gemv_f16. */
int sp_gemv_batched_f16(const void *x, int64_t ldx, int64_t x_stride, const void *w, int64_t w_stride,
                        const float *b, int64_t b_stride, float *y, void *y_f16, int64_t ldy, int64_t y_stride,
                        int batch, int rows, int n, int k, int silu_in, int silu_out, void *stream);

/* Sinusoidal embedding [cos | sin] (diffusers Timesteps(dim, flip_sin_to_cos=True, shift=0)) of
 * `count` fp32 values read from device memory; writes fp16 [count][dim]. */
int sp_sinusoid_f16(const float *values, void *out, int count, int dim, void *stream);

/* ---------------------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU), channels-last.  Replaces torch.nn.GroupNorm + F.silu in
 * ResnetBlock2D / TemporalResnetBlock / TransformerSpatioTemporalModel.norm / conv_norm_out.
 * x,y: fp16 [instances][rows][C]; statistics are taken per (instance, group) over rows x C/groups.
 * Spatial norm: instance = frame; temporal norm: instance = batch (rows = frames*H*W).
 * ws: >= sp_groupnorm_ws_bytes() bytes of scratch.
 * ------------------------------------------------------------------------------------------- */
size_t sp_groupnorm_ws_bytes(int instances, int64_t rows, int c, int groups);
int sp_groupnorm_f16(const void *x, const float *gamma, const float *beta, void *y, int instances,
                     int64_t rows, int c, int groups, float eps, int fuse_silu, void *ws,
                     size_t ws_bytes, void *stream);
/* The same with x a column slice of a wider row-major tensor: rows of ldx halves (ldx >= C, a multiple of 8), the C
 * channels at the head of each; y stays dense.  (The skip tensors of the down path live inside the buffers the up path
 * would otherwise build with torch.cat -- unet_spatio_temporal_condition.py up blocks -- so the norms that read them
 * in the down path see strided rows.) */
int sp_groupnorm_ld_f16(const void *x, int64_t ldx, const float *gamma, const float *beta, void *y, int instances,
                        int64_t rows, int c, int groups, float eps, int fuse_silu, void *ws,
                        size_t ws_bytes, void *stream);

/* GroupNorm(+SiLU) of a tensor whose producer left per-tile column sums (sp_gemm_desc.gn_part, `part` = that buffer for
 * the [instances*rows][c] tensor x with c = the producer's n): one small kernel folds the sums of every instance's tiles
 * into (mean, rstd) per (instance, group) in fp64 (fixed order), then the apply pass of sp_groupnorm_f16 runs -- no
 * statistics pass over x.  rows must be a multiple of 256 (an instance = whole tiles).  stats: fp32 scratch of
 * instances*groups*2 floats. */
int sp_groupnorm_tile_sums_f16(const void *x, int64_t ldx, const float *part, const float *gamma, const float *beta, void *y,
                               int instances, int64_t rows, int c, int groups, float eps, int fuse_silu, float *stats,
                               void *stream);
/* The same for a tensor that is the CONCATENATION [a | b] of two producers' outputs (an up block's resnet normalises
 * [hidden | skip]): channels [0, c_a) are summed in `part` (row pitch c_a), channels [c_a, c) in `part_b` (row pitch c - c_a);
 * per-column sums are additive, so a group may straddle the seam. */
int sp_groupnorm_tile_sums2_f16(const void *x, int64_t ldx, const float *part, int c_a, const float *part_b, const float *gamma,
                                const float *beta, void *y, int instances, int64_t rows, int c, int groups, float eps,
                                int fuse_silu, float *stats, void *stream);
/* sp_groupnorm_fold_linear_f16 (below) with the statistics folded from such column sums instead of a pass over x. */
int sp_groupnorm_fold_linear_tile_sums_f16(const float *part, const float *gamma, const float *beta, int instances, int64_t rows,
                                           int c, int groups, float eps, const void *w, const float *bias, int n, void *w_out,
                                           float *bias_out, float *stats, void *stream);
/* GroupNorm (no activation) folded into the nn.Linear that consumes it -- diffusers TransformerSpatioTemporalModel:
 * hidden = proj_in(norm(x)) -- so that the normalised tensor is never written or read:
 *   GN(x)[r][c] = (x[r][c] - mean[i][g(c)]) * rstd[i][g(c)] * gamma[c] + beta[c]      (i = instance of row r)
 *   proj_in(GN(x))[r][n] = sum_c x[r][c] * w_out[i][n][c] + bias_out[i][n]
 *   w_out[i][n][c]  = fp16( w[n][c] * gamma[c] * rstd[i][g(c)] )
 *   bias_out[i][n]  = bias[n] + sum_c beta[c]*w[n][c] - sum_c mean[i][g(c)] * float(w_out[i][n][c])
 * (the mean term uses the ROUNDED weights, so it cancels exactly what the MFMA adds for a constant offset of the group).
 * One statistics pass over x (the same kernels as sp_groupnorm_f16's first pass) + one small kernel that writes the
 * instances * n * c scaled weights; the caller then runs sp_gemm_f16 on the RAW x with w = w_out, w_group_rows = rows,
 * w_group_stride = n*c, bias = NULL, bias2 = bias_out, bias2_rows = rows.  x: fp16 [instances*rows][ldx] (C channels at
 * the head of each row); w: fp16 [n][c]; gamma, beta, bias: fp32 (bias may be NULL); ws as for sp_groupnorm_f16. */
int sp_groupnorm_fold_linear_f16(const void *x, int64_t ldx, const float *gamma, const float *beta, int instances,
                                 int64_t rows, int c, int groups, float eps, const void *w, const float *bias, int n,
                                 void *w_out, float *bias_out, void *ws, size_t ws_bytes, void *stream);

/* LayerNorm over the last dim (torch.nn.LayerNorm in BasicTransformerBlock /
 * TemporalBasicTransformerBlock).  Optional pre-add of a per-frame vector
 * (`hidden_states_mix = hidden_states + emb`): xin = x + addvec[row / addvec_rows]; if `sum_out`
 * is non-NULL the pre-norm sum is stored there as well.  fp16 in/out, fp32 affine. */
int sp_layernorm_f16(const void *x, const void *addvec, int64_t addvec_rows, void *sum_out,
                     const float *gamma, const float *beta, void *y, int64_t rows, int c, float eps,
                     void *stream);
/* Statistics only, for a LayerNorm that is folded into the next GEMM (sp_gemm_desc.ln_stats): stats[row] = (mean,
 * 1/sqrt(var + eps)) over the C channels of xin (same optional pre-add / sum_out as sp_layernorm_f16).  One read of
 * x instead of a read and a write. */
int sp_ln_stats_f16(const void *x, const void *addvec, int64_t addvec_rows, void *sum_out, float *stats,
                    int64_t rows, int c, float eps, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Attention.  Replaces F.scaled_dot_product_attention / xformers (svd_unet.py:142) for
 *  - spatial self-attention: sequences of `seq` tokens, `batch` of them (one per frame),
 *    q,k,v,o fp16 [batch*seq][heads*64] with row stride ld* (so q,k,v may alias one fused QKV buffer)
 *  - temporal self-attention: sequences run ACROSS frames for every pixel; token (f, p) lives at
 *    row f*hw + p, so no permute is materialised.
 * head_dim is fixed at 64 (all SVD levels).  softmax scale = 1/8.  zero_page: SP_ZERO_PAGE_BYTES of
 * zero-filled device memory (padding rows are read from it).
 * ------------------------------------------------------------------------------------------- */
int sp_attn_spatial_f16(const void *q, const void *k, const void *v, void *o, int64_t ldq,
                        int64_t ldk, int64_t ldv, int64_t ldo, int batch, int seq, int heads,
                        float scale, const void *zero_page, void *stream);
int sp_attn_temporal_f16(const void *q, const void *k, const void *v, void *o, int64_t ldq,
                         int64_t ldk, int64_t ldv, int64_t ldo, int batch, int frames, int64_t hw,
                         int heads, float scale, const void *zero_page, void *stream);
/* sp_attn_spatial_f16 for LONG rows (csrc/attention_long.hip): same arguments, same results to fp16 rounding.
 * One wave per SIMD with two query blocks; after a warm-up over the workgroup's own tokens and tile 0 (ordinary
 * online softmax) each row's reference is frozen at the maximum seen so far and the remaining K/V tiles run without
 * row maximum or rescale.  A later score more than 16 (log2 units) above that reference overflows fp16, is detected
 * through the row sum, and the 256-row block is recomputed by sp_attn_spatial_f16's kernel inside the same call, so
 * accuracy never depends on the data (only the time does).  Applies to seq >= 4096 with seq % 256 == 0; every other
 * shape is passed to sp_attn_spatial_f16's kernel unchanged (workspace unused).  workspace: >=
 * sp_attn_long_ws_bytes(batch, seq, heads) bytes, 4-byte aligned, owned by the caller and private to the call until
 * it has completed on `stream` (one flag word per 256 query rows; zeroed by the call). */
int64_t sp_attn_long_ws_bytes(int batch, int seq, int heads);
int sp_attn_spatial_long_f16(const void *q, const void *k, const void *v, void *o, int64_t ldq, int64_t ldk,
                             int64_t ldv, int64_t ldo, int batch, int seq, int heads, float scale,
                             const void *zero_page, void *workspace, int64_t workspace_bytes, void *stream);
/* fp8 (OCP e4m3fn) MFMA variant of sp_attn_spatial_f16 for BASELINE config 5 ("fp8 MFMA attention path"; the
 * reference has no fp8 path of its own, its attention is diffusers/xformers behind svd_unet.py:142-199).
 * Same arguments and fp16 inputs/outputs; q/k/v are quantised per call into `workspace` (Q8, K8 row-major,
 * V8 transposed per head) and S^T = K.Q^T, O^T += V^T.P^T run on v_mfma_f32_32x32x16_fp8_fp8 with fp32
 * accumulation and fp32 softmax.  workspace: >= sp_attn_fp8_ws_bytes(batch, seq, heads) bytes, 16-byte aligned,
 * owned by the caller (never allocated here).  Tolerance vs the fp32 oracle: rel-L2 <= 3e-2. */
int64_t sp_attn_fp8_ws_bytes(int batch, int seq, int heads);
int sp_attn_spatial_fp8(const void *q, const void *k, const void *v, void *o, int64_t ldq, int64_t ldk,
                        int64_t ldv, int64_t ldo, int batch, int seq, int heads, float scale,
                        void *workspace, int64_t workspace_bytes, const void *zero_page, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Layout / elementwise glue around the UNet (svd_unet.py:382-439).
 * ------------------------------------------------------------------------------------------- */
/* latent (B,4,F,H,W) * in_scale  ++  image_latents (B,4,F,H,W)  ->  NHWC rows [B*F*H*W][cpad]
 * (channels 0-3 scaled latent, 4-7 image latents, rest zero): svd_unet.py:382,414-415 */
int sp_pack_input_f16(const void *latent, const void *image_latents, void *out, float in_scale,
                      int b, int frames, int h, int w, int cpad, void *stream);
/* v-prediction Euler update with optional CFG, fp32 math (svd_unet.py:410-411,425-439):
 * eps = eps_u + gs[f]*(eps_c - eps_u) if eps_uncond!=NULL else eps_c;
 * x' = x + ((x - (eps*c_out + x*c_skip))/sigma)*dt.  eps_* are NHWC [B*F*H*W][ld_eps]; latent and
 * out are (B,4,F,H,W). */
int sp_euler_step_f16(const void *latent, const void *eps_cond, const void *eps_uncond,
                      int64_t ld_eps, const float *guidance /*[F] or NULL*/, void *out, float sigma,
                      float sigma_next, int b, int frames, int h, int w, void *stream);
/* channel concat of two NHWC tensors (torch.cat([hidden, skip], dim=1) in the up blocks) */
int sp_concat_channels_f16(const void *a, int ca, const void *b, int cb, void *out, int64_t rows,
                           void *stream);
/* y = x + vec[c] broadcast over rows (used for the degenerate single-token cross-attention) */
int sp_add_rowvec_f16(const void *x, const float *vec, void *y, int64_t rows, int c, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Temporal VAE decoder on the last stage: /root/reference/scripts/generate_video_demo.py:154-195
 * calls diffusers AutoencoderKLTemporalDecoder.decode; its convolutions, GroupNorms and projections are the kernels
 * above, the two below are what it needs besides.
 * ------------------------------------------------------------------------------------------- */
/* in-place softmax over each of `rows` rows of x (fp16 [rows][ld], `cols` <= 16384 columns, multiple of 8): the
 * [tokens][tokens] scores of the mid block's single-head attention (attention_processor.py Attention, heads = 1,
 * dim_head = 512), which sp_gemm_f16 wrote already scaled by 1/sqrt(dim_head).  fp32 statistics. */
int sp_softmax_rows_f16(void *x, int64_t ld, int64_t rows, int cols, void *stream);
/* The same attention with logits that never pass through fp16 (the reference runs this VAE in fp32 because trained
 * weights overflow fp16: scripts/generate_video_demo.py:171-175):
 *   sp_gemm_f32out_f16: D[m][n] (fp32, row pitch n) = A[m][k] . W[n][k]^T, raw fp32 sums, no bias / epilogue
 *                       (a fp16 [m][lda], w fp16 [n][k]; n a multiple of 256, k a multiple of 64, d 16-byte aligned);
 *   sp_softmax_rows_f32: out[r][c] = fp16(softmax_c(scale * x[r][c])), x fp32 [rows][ld], out fp16 [rows][ldo], fp32
 *                       statistics, nothing clamped (a NaN stays a NaN).  out may be x itself with ldo = 2*ld: the
 *                       probabilities of a row then overwrite the front of that row's logits (no second matrix). */
int sp_gemm_f32out_f16(const void *a, int64_t lda, const void *w, void *d, int m, int n, int k, const void *zero_page,
                       void *stream);
int sp_softmax_rows_f32(const float *x, int64_t ld, void *out, int64_t ldo, int64_t rows, int cols, float scale, void *stream);
/* Both ends of one decoder call work on `n` consecutive entries g = flat0 .. flat0+n-1 of the flattened (batch, frame)
 * list of a video tensor with F frames per batch item (generate_video_demo.py:162-181 cuts that flat list into chunks
 * of decode_chunk_size); entry g is batch item g / F, frame g % F, and element (g, channel c, pixel p) of the tensor
 * sits at  base + (g/F)*sb + c*sc + (g%F)*sf + p  (element strides: (B,C,F,H,W) has sb = C*F*hw, sc = F*hw, sf = hw;
 * (B*F,C,H,W) has sb = F*C*hw, sc = hw, sf = C*hw).
 * pack: latent channels 0-3 * scale (= 1/scaling_factor) -> channels-last rows [n*h*w][cpad], other channels zero. */
int sp_vae_pack_latent_f16(const void *latent, void *rows, float scale, int64_t flat0, int n, int F, int64_t sb,
                           int64_t sc, int64_t sf, int h, int w, int cpad, void *stream);
/* frames out: time_conv_out = Conv3d(3 -> 3, kernel (3,1,1), zero padding over the `frames` of each of the call's
 * `batch` items, n = batch*frames) applied to the channels-last rows conv_out produced (fp16 [n*h*w][ld], channels
 * 0-2), written into the video tensor `out` (fp16, or fp32 when out_fp32 != 0) at the strides above.
 * weight: fp32 [3][3][3] = [out][in][tap], bias fp32 [3]. */
int sp_vae_frames_out_f16(const void *rows, int64_t ld, const float *weight, const float *bias, void *out,
                          int out_fp32, int batch, int frames, int h, int w, int64_t flat0, int F, int64_t sb,
                          int64_t sc, int64_t sf, void *stream);

/* Encoder half (vae.encode(image).latent_dist.mode(), generate_video_demo.py:139-148).  flip != 0 mirrors the image in
 * both axes between the tensor and the rows (pixel (y,x) <-> row (H-1-y)*W + (W-1-x)): the engine runs the encoder on
 * the mirrored image, where Downsample2D's bottom/right padding is the stride-2 kernel's top/left padding.
 * pack: image fp16 (batch,3,h,w) -> channels-last rows [batch*h*w][cpad] (channels 3.. zero). */
int sp_vae_image_pack_f16(const void *image, void *rows, int batch, int h, int w, int cpad, int flip, void *stream);
/* latent out: rows fp16 [batch*h*w][ld], channels 0..channels-1 -> out fp16 (batch, channels, frames, h, w), every frame
 * a copy (image_latents.unsqueeze(2).repeat(1,1,num_frames,1,1), generate_video_demo.py:148). */
int sp_vae_latent_out_f16(const void *rows, int64_t ld, void *out, int batch, int channels, int frames, int h, int w,
                          int flip, void *stream);

/* ---------------------------------------------------------------------------------------------
 * CLIP image encoder on the first stage: /root/reference/scripts/generate_video_demo.py:108-112 calls
 * transformers CLIPVisionModelWithProjection (ViT-H/14); its contractions and LayerNorms are the kernels above.
 * ------------------------------------------------------------------------------------------- */
/* im2col of non-overlapping patches: pixels fp16 (batch,3,h,w) -> rows fp16 [batch*(h/patch)*(w/patch)][kpad],
 * k = c*patch*patch + ky*patch + kx (= the flattened Conv2d weight of CLIPVisionEmbeddings.patch_embedding), zeros
 * from 3*patch*patch up to kpad. */
int sp_patchify_f16(const void *pixels, void *rows, int batch, int h, int w, int patch, int kpad, void *stream);
/* softmax(q k^T * scale) v for short sequences (seq <= 512) and any head width that is a multiple of 8 up to 128
 * (ViT-H: 257 tokens, 16 heads of 80): row (b*seq + i), head hh of q/k/v/o starts at column hh*head_dim.  fp32 math,
 * fp16 in/out; K and V of one (batch item, head) must fit in LDS. */
int sp_attn_small_f16(const void *q, const void *k, const void *v, void *o, int64_t ldq, int64_t ldk, int64_t ldv,
                      int64_t ldo, int batch, int seq, int heads, int head_dim, float scale, void *stream);
/* y = gelu(x) elementwise over n fp16 values (n a multiple of 8): exact erf form (hidden_act "gelu"), or
 * x*sigmoid(1.702 x) when quick != 0 ("quick_gelu"). */
int sp_gelu_f16(const void *x, void *y, int64_t n, int quick, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement aid (bench.py `roofline.clock_ghz_live`; the reference has no counterpart -- its benchmark reads no clocks,
 * /root/reference/src/modes/benchmark.py:170-262).  One time stamp in stream order: `blocks` one-wave workgroups each write
 * four u64 words to out[block][4]: the id of the XCD the workgroup ran on (HW_REG_XCC_ID), the shader-clock counter
 * (s_memtime), the constant 100 MHz counter (s_memrealtime), and 1.  Two stamps taken on the same XCD before and after a
 * stretch of work give the shader clock held in between: GHz = 0.1 * (shader ticks) / (100-MHz ticks).
 * ------------------------------------------------------------------------------------------- */
int sp_clock_stamp(void *out, int blocks, void *stream);

/* ---------------------------------------------------------------------------------------------
 * DummyUNet (simulator-path model, /root/reference/src/models/dummy_unet.py:37-59), fp32 NCDHW:
 * out = x + gain*Conv3d(SiLU(Conv3d(x))) + LayerNorm_C(x).  hidden: scratch [B][hidden][F][H][W].
 * ------------------------------------------------------------------------------------------- */
int sp_dummy_unet_f32(const float *x, float *out, float *hidden, const float *w1, const float *b1,
                      const float *w2, const float *b2, const float *ln_w, const float *ln_b,
                      float ln_eps, int use_ln, float gain, int b, int c, int hidden_c, int frames,
                      int h, int w, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SVDPIPE_H */

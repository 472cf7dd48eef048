// Kernels only the temporal VAE decoder needs (/root/reference/scripts/generate_video_demo.py:154-195 ->
// diffusers AutoencoderKLTemporalDecoder.decode); its convolutions, (3,1,1) convolutions, GroupNorms and projections
// are the UNet's kernels (gemm*.hip, norm.hip).
//
//   softmax_rows_kernel : in-place row softmax of the [tokens][tokens] score matrix of the mid block's single-head,
//                         512-wide attention (scores come out of sp_gemm_f16 already scaled by 1/sqrt(C)).  One
//                         workgroup per row, the row lives in registers (packed fp16), fp32 statistics.  HBM-bound:
//                         one read + one write of the matrix.
//   pack_latent_kernel  : a chunk of latent frames / scaling_factor -> channels-last rows (the decoder's input).
//   frames_out_kernel   : time_conv_out -- Conv3d(3 -> 3, kernel (3,1,1), zero padding in time) -- on the channels-last
//                         rows conv_out wrote, stored straight into the video tensor the caller returns.  HBM-bound.
#include "common.h"

namespace {

// A score that overflowed fp16 on its way out of the contraction (+-inf) is read as +-65504: the row then has a finite
// maximum and no inf - inf = NaN; a single saturated score gets (correctly) all of the row's mass.
// A NaN stays a NaN (an upstream inf - inf must remain visible to decode_latents(check_finite=True)).
__device__ __forceinline__ float finite_f16(f16 h) {
  const float f = (float)h;
  return f != f ? f : __builtin_amdgcn_fmed3f(f, -65504.0f, 65504.0f);
}

// Row softmax of fp32 logits: p = softmax(scale * s) written as fp16.  One workgroup per row, the row in registers (fp32),
// fp32 statistics; `out` may be the FRONT of the same row (in-place shrink: every thread holds its logits in registers
// before the first probability is written -- the two barriers of the reductions lie between).  Nothing is clamped: the
// logits never passed through fp16, and a NaN stays a NaN.
template <int MAXV>
__global__ __launch_bounds__(256) void softmax_rows_f32_kernel(const float *__restrict__ x, int64_t ld, f16 *__restrict__ out,
                                                               int64_t ldo, int cols, float scale_log2e) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *row = x + (int64_t)blockIdx.x * ld;
  f16 *orow = out + (int64_t)blockIdx.x * ldo;
  const int oc = cols >> 2;                       // 16-byte pieces of four logits
  f32x4 v[MAXV];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int o = tid + i * 256;
    if (o < oc) {
      v[i] = *(const f32x4 *)(row + o * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) mx = fmaxf(mx, v[i][e]);
    }
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float nm = -mx * scale_log2e;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int o = tid + i * 256;
    if (o < oc) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[i][e] = __builtin_amdgcn_exp2f(fmaf(v[i][e], scale_log2e, nm)); s += v[i][e]; }
    }
  }
  s = wave_sum(s);
  if (lane == 0) red[4 + wave] = s;
  __syncthreads();
  const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));     // fixed order: deterministic
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int o = tid + i * 256;
    if (o < oc) {
      const f16x4 w = {(f16)(v[i][0] * inv), (f16)(v[i][1] * inv), (f16)(v[i][2] * inv), (f16)(v[i][3] * inv)};
      *(f16x4 *)(orow + o * 4) = w;
    }
  }
}

template <int MAXV>
__global__ __launch_bounds__(256) void softmax_rows_kernel(f16 *__restrict__ x, int64_t ld, int cols) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f16 *row = x + (int64_t)blockIdx.x * ld;
  const int oc = cols >> 3;
  f16x8 v[MAXV];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int o = tid + i * 256;
    if (o < oc) {
      v[i] = *(const f16x8 *)(row + o * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, finite_f16(v[i][e]));
    }
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  // p = exp(x - max): kept in fp32 only long enough to be summed; the row is re-derived from the packed values below
  float s = 0.f;
  const float nm = -mx * 1.4426950408889634f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int o = tid + i * 256;
    if (o < oc) {
#pragma unroll
      for (int e = 0; e < 8; ++e) s += __builtin_amdgcn_exp2f(fmaf(finite_f16(v[i][e]), 1.4426950408889634f, nm));
    }
  }
  s = wave_sum(s);
  if (lane == 0) red[4 + wave] = s;
  __syncthreads();
  const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));     // fixed order: deterministic
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int o = tid + i * 256;
    if (o < oc) {
      f16x8 w;
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = (f16)(__builtin_amdgcn_exp2f(fmaf(finite_f16(v[i][e]), 1.4426950408889634f, nm)) * inv);
      *(f16x8 *)(row + o * 8) = w;
    }
  }
}

// one thread per (entry, pixel) of the call
__global__ void pack_latent_kernel(const f16 *__restrict__ lat, f16 *__restrict__ out, float scale, int64_t flat0,
                                   int F, int64_t sb, int64_t sc, int64_t sf, int64_t hw, int cpad, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t p = idx % hw, g = flat0 + idx / hw;
  const f16 *src = lat + (g / F) * sb + (g % F) * sf + p;
  f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = (f16)((float)src[c * sc] * scale);
  f16 *o = out + idx * cpad;
  *(f16x8 *)o = v;
  const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int c = 8; c < cpad; c += 8) *(f16x8 *)(o + c) = z;
}

// one thread per (frame, pixel): out[g][co][p] = b[co] + sum_{tap, ci} w[co][ci][tap] * in[f + tap - 1][p][ci]
template <typename OUT>
__global__ void frames_out_kernel(const f16 *__restrict__ rows, int64_t ld, const float *__restrict__ w,
                                  const float *__restrict__ b, OUT *__restrict__ out, int frames, int64_t hw,
                                  int64_t total, int64_t flat0, int F, int64_t sb, int64_t sc, int64_t sf) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t p = idx % hw, bf = idx / hw;
  const int f = (int)(bf % frames);
  float acc[3] = {b[0], b[1], b[2]};
#pragma unroll
  for (int tap = 0; tap < 3; ++tap) {
    const int ff = f + tap - 1;
    if (ff < 0 || ff >= frames) continue;
    const f16x4 v = *(const f16x4 *)(rows + (idx + (int64_t)(tap - 1) * hw) * ld);
#pragma unroll
    for (int co = 0; co < 3; ++co)
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) acc[co] = fmaf(w[(co * 3 + ci) * 3 + tap], (float)v[ci], acc[co]);
  }
  const int64_t g = flat0 + bf;
  OUT *dst = out + (g / F) * sb + (g % F) * sf + p;
#pragma unroll
  for (int co = 0; co < 3; ++co) dst[co * sc] = (OUT)acc[co];
}

// Encoder ends.  `flip`: rows hold the image mirrored in both axes (pixel (y, x) at row (H-1-y)*W + (W-1-x)); see
// models/vae_hip.py::ImageEncoderHIP for why the encoder runs on the mirrored image.
__global__ void image_pack_kernel(const f16 *__restrict__ img, f16 *__restrict__ out, int h, int w, int cpad, int flip,
                                  int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;    // one output row
  if (idx >= total) return;
  const int64_t hw = (int64_t)h * w, p = idx % hw, b = idx / hw;
  const int64_t src = flip ? hw - 1 - p : p;
  f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < 3; ++c) v[c] = img[(b * 3 + c) * hw + src];
  f16 *o = out + idx * cpad;
  *(f16x8 *)o = v;
  const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int c = 8; c < cpad; c += 8) *(f16x8 *)(o + c) = z;
}

// rows [b*hw][ld] channels 0..c-1  ->  out (b, c, frames, h, w), every frame the same image latent
__global__ void latent_out_kernel(const f16 *__restrict__ rows, int64_t ld, f16 *__restrict__ out, int c, int frames,
                                  int64_t hw, int flip, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;    // one (b, pixel)
  if (idx >= total) return;
  const int64_t p = idx % hw, b = idx / hw;
  const int64_t src = b * hw + (flip ? hw - 1 - p : p);
  for (int ch = 0; ch < c; ++ch) {
    const f16 v = rows[src * ld + ch];
    for (int f = 0; f < frames; ++f) out[((b * c + ch) * frames + f) * hw + p] = v;
  }
}

}  // namespace

extern "C" int sp_softmax_rows_f16(void *x, int64_t ld, int64_t rows, int cols, void *stream) {
  SP_REQUIRE(x, "sp_softmax_rows_f16: null pointer");
  SP_REQUIRE(rows > 0 && rows <= 0x7fffffff && cols >= 8 && cols % 8 == 0 && cols <= 8 * 256 * 8 && ld >= cols && ld % 8 == 0,
             "sp_softmax_rows_f16: rows=%lld cols=%d ld=%lld unsupported (cols a multiple of 8, <= 16384)",
             (long long)rows, cols, (long long)ld);
  hipStream_t s = (hipStream_t)stream;
  const int oc = cols / 8;
  SP_CLEAR_STALE_ERROR();
  if (oc <= 256)
    hipLaunchKernelGGL(softmax_rows_kernel<1>, dim3((unsigned)rows), dim3(256), 0, s, (f16 *)x, ld, cols);
  else if (oc <= 512)
    hipLaunchKernelGGL(softmax_rows_kernel<2>, dim3((unsigned)rows), dim3(256), 0, s, (f16 *)x, ld, cols);
  else if (oc <= 1280)
    hipLaunchKernelGGL(softmax_rows_kernel<5>, dim3((unsigned)rows), dim3(256), 0, s, (f16 *)x, ld, cols);
  else
    hipLaunchKernelGGL(softmax_rows_kernel<8>, dim3((unsigned)rows), dim3(256), 0, s, (f16 *)x, ld, cols);
  SP_CHECK_LAUNCH("sp_softmax_rows_f16");
  return SP_OK;
}

extern "C" int sp_softmax_rows_f32(const float *x, int64_t ld, void *out, int64_t ldo, int64_t rows, int cols, float scale,
                                   void *stream) {
  SP_REQUIRE(x && out, "sp_softmax_rows_f32: null pointer");
  SP_REQUIRE(rows > 0 && rows <= 0x7fffffff && cols >= 4 && cols % 4 == 0 && cols <= 4 * 256 * 16 && ld >= cols && ld % 4 == 0 &&
                 ldo >= cols && ldo % 4 == 0,
             "sp_softmax_rows_f32: rows=%lld cols=%d ld=%lld ldo=%lld unsupported (cols a multiple of 4, <= 16384)",
             (long long)rows, cols, (long long)ld, (long long)ldo);
  // in place = the probabilities of row r overwrite the front of row r's logits: rows must not overlap other rows' logits
  SP_REQUIRE((const void *)x != (const void *)out || ldo == 2 * ld,
             "sp_softmax_rows_f32: in place needs ldo == 2 * ld (fp16 row pitch in halves = fp32 row pitch in bytes / 2)");
  hipStream_t s = (hipStream_t)stream;
  const int oc = cols / 4;
  const float sl = scale * 1.4426950408889634f;
  SP_CLEAR_STALE_ERROR();
  if (oc <= 256 * 2)
    hipLaunchKernelGGL(softmax_rows_f32_kernel<2>, dim3((unsigned)rows), dim3(256), 0, s, x, ld, (f16 *)out, ldo, cols, sl);
  else if (oc <= 256 * 5)
    hipLaunchKernelGGL(softmax_rows_f32_kernel<5>, dim3((unsigned)rows), dim3(256), 0, s, x, ld, (f16 *)out, ldo, cols, sl);
  else if (oc <= 256 * 9)
    hipLaunchKernelGGL(softmax_rows_f32_kernel<9>, dim3((unsigned)rows), dim3(256), 0, s, x, ld, (f16 *)out, ldo, cols, sl);
  else
    hipLaunchKernelGGL(softmax_rows_f32_kernel<16>, dim3((unsigned)rows), dim3(256), 0, s, x, ld, (f16 *)out, ldo, cols, sl);
  SP_CHECK_LAUNCH("sp_softmax_rows_f32");
  return SP_OK;
}

extern "C" int sp_vae_pack_latent_f16(const void *latent, void *rows, float scale, int64_t flat0, int n, int F,
                                      int64_t sb, int64_t sc, int64_t sf, int h, int w, int cpad, void *stream) {
  SP_REQUIRE(latent && rows, "sp_vae_pack_latent_f16: null pointer");
  SP_REQUIRE(flat0 >= 0 && n > 0 && F > 0 && h > 0 && w > 0 && sb > 0 && sc > 0 && sf > 0,
             "sp_vae_pack_latent_f16: dims and strides must be positive");
  SP_REQUIRE(cpad >= 8 && cpad % 8 == 0, "sp_vae_pack_latent_f16: cpad=%d must be a multiple of 8, >= 8", cpad);
  const int64_t hw = (int64_t)h * w, total = (int64_t)n * hw;
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(pack_latent_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const f16 *)latent, (f16 *)rows, scale, flat0, F, sb, sc, sf, hw, cpad, total);
  SP_CHECK_LAUNCH("sp_vae_pack_latent_f16");
  return SP_OK;
}

extern "C" int sp_vae_frames_out_f16(const void *rows, int64_t ld, const float *weight, const float *bias, void *out,
                                     int out_fp32, int batch, int frames, int h, int w, int64_t flat0, int F,
                                     int64_t sb, int64_t sc, int64_t sf, void *stream) {
  SP_REQUIRE(rows && weight && bias && out, "sp_vae_frames_out_f16: null pointer");
  SP_REQUIRE(batch > 0 && frames > 0 && h > 0 && w > 0 && ld >= 4 && ld % 4 == 0,
             "sp_vae_frames_out_f16: bad shape (ld=%lld must be a multiple of 4, >= 4)", (long long)ld);
  SP_REQUIRE(flat0 >= 0 && F > 0 && sb > 0 && sc > 0 && sf > 0, "sp_vae_frames_out_f16: F and strides must be positive");
  const int64_t hw = (int64_t)h * w, total = (int64_t)batch * frames * hw;
  SP_REQUIRE((total + 255) / 256 <= 0x7fffffff, "sp_vae_frames_out_f16: too many pixels");
  SP_CLEAR_STALE_ERROR();
  if (out_fp32)
    hipLaunchKernelGGL(frames_out_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const f16 *)rows, ld, weight, bias, (float *)out, frames, hw, total, flat0, F, sb, sc, sf);
  else
    hipLaunchKernelGGL(frames_out_kernel<f16>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const f16 *)rows, ld, weight, bias, (f16 *)out, frames, hw, total, flat0, F, sb, sc, sf);
  SP_CHECK_LAUNCH("sp_vae_frames_out_f16");
  return SP_OK;
}

extern "C" int sp_vae_image_pack_f16(const void *image, void *rows, int batch, int h, int w, int cpad, int flip,
                                     void *stream) {
  SP_REQUIRE(image && rows, "sp_vae_image_pack_f16: null pointer");
  SP_REQUIRE(batch > 0 && h > 0 && w > 0 && cpad >= 8 && cpad % 8 == 0, "sp_vae_image_pack_f16: bad shape / cpad=%d", cpad);
  const int64_t total = (int64_t)batch * h * w;
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(image_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const f16 *)image, (f16 *)rows, h, w, cpad, flip, total);
  SP_CHECK_LAUNCH("sp_vae_image_pack_f16");
  return SP_OK;
}

extern "C" int sp_vae_latent_out_f16(const void *rows, int64_t ld, void *out, int batch, int channels, int frames, int h,
                                     int w, int flip, void *stream) {
  SP_REQUIRE(rows && out, "sp_vae_latent_out_f16: null pointer");
  SP_REQUIRE(batch > 0 && channels > 0 && frames > 0 && h > 0 && w > 0 && ld >= channels,
             "sp_vae_latent_out_f16: bad shape (ld=%lld, channels=%d)", (long long)ld, channels);
  const int64_t hw = (int64_t)h * w, total = (int64_t)batch * hw;
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(latent_out_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const f16 *)rows, ld, (f16 *)out, channels, frames, hw, flip, total);
  SP_CHECK_LAUNCH("sp_vae_latent_out_f16");
  return SP_OK;
}

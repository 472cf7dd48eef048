// Layout / elementwise glue around the UNet forward and the small M=1 GEMVs of the embedding path.
#include "common.h"

namespace {

// latent (B,4,F,H,W)*scale ++ image_latents (B,4,F,H,W) -> NHWC rows [(b,f,y,x)][cpad]
__global__ void pack_input_kernel(const f16 *__restrict__ lat, const f16 *__restrict__ img,
                                  f16 *__restrict__ out, float scale, int frames, int64_t hw, int cpad,
                                  int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one output row
  if (idx >= total) return;
  const int64_t p = idx % hw;
  const int64_t bf = idx / hw;
  const int f = (int)(bf % frames);
  const int64_t b = bf / frames;
  f16x8 v;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int64_t src = ((b * 4 + c) * frames + f) * hw + p;
    // reference: latent / sqrt(sigma^2+1) evaluated in the storage dtype (svd_unet.py:382)
    v[c] = (f16)((float)lat[src] * scale);
    v[c + 4] = img[src];
  }
  f16 *o = out + idx * cpad;
  *(f16x8 *)o = v;
  const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int c = 8; c < cpad; c += 8) *(f16x8 *)(o + c) = z;
}

__global__ void euler_kernel(const f16 *__restrict__ lat, const f16 *__restrict__ ec,
                             const f16 *__restrict__ eu, int64_t ld_eps, const float *__restrict__ gs,
                             f16 *__restrict__ out, float c_out, float c_skip, float inv_sigma, float dt,
                             int frames, int64_t hw, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one (b,f,pixel)
  if (idx >= total) return;
  const int64_t p = idx % hw;
  const int64_t bf = idx / hw;
  const int f = (int)(bf % frames);
  const int64_t b = bf / frames;
  const f16x4 c4 = *(const f16x4 *)(ec + idx * ld_eps);
  f16x4 u4 = c4;
  float g = 1.f;
  if (eu) { u4 = *(const f16x4 *)(eu + idx * ld_eps); g = gs[f]; }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int64_t a = ((b * 4 + c) * frames + f) * hw + p;
    const float x = (float)lat[a];
    float e = (float)c4[c];
    if (eu) {
      // reference evaluates the guidance mix in fp16 (svd_unet.py:410-411)
      const f16 gh = (f16)g;
      const f16 diff = (f16)((float)c4[c] - (float)u4[c]);
      const f16 prod = (f16)((float)gh * (float)diff);
      e = (float)(f16)((float)u4[c] + (float)prod);
    }
    const float x0 = e * c_out + x * c_skip;
    const float d = (x - x0) * inv_sigma;
    out[a] = (f16)(x + d * dt);
  }
}

__global__ void concat_kernel(const f16 *__restrict__ a, int oa, const f16 *__restrict__ b, int ob,
                              f16 *__restrict__ out, int64_t total) {
  // one thread per output octet
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int oc = oa + ob;
  const int64_t row = idx / oc;
  const int o = (int)(idx - row * oc);
  const f16x8 v = o < oa ? *(const f16x8 *)(a + (row * oa + o) * 8)
                         : *(const f16x8 *)(b + (row * ob + (o - oa)) * 8);
  *(f16x8 *)(out + idx * 8) = v;
}

__global__ void add_rowvec_kernel(const f16 *__restrict__ x, const float *__restrict__ vec,
                                  f16 *__restrict__ y, int oc, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int o = (int)(idx % oc);
  f16x8 v = *(const f16x8 *)(x + idx * 8);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (f16)((float)v[e] + vec[o * 8 + e]);
  *(f16x8 *)(y + idx * 8) = v;
}

// y[row][n] = W[n][:] . act(x[row][:]) + b[n].  A wave owns RC consecutive output elements and has the RC weight rows
// of a K slice in flight together (one output per wave kept a single 16-byte load per lane in flight: 0.3 TB/s on the
// 77 MB of time-embedding projections).  blockIdx.z = problem of a batch of same-shape GEMVs (strides in elements;
// 0 = shared operand)
constexpr int GEMV_RC = 8;
__global__ __launch_bounds__(256) void gemv_kernel(const f16 *__restrict__ x, int64_t ldx,
                                                   const f16 *__restrict__ w, const float *__restrict__ b,
                                                   float *__restrict__ y, f16 *__restrict__ yh, int64_t ldy,
                                                   int n, int k, int silu_in, int silu_out, int64_t xs,
                                                   int64_t ws, int64_t bs, int64_t ys) {
  constexpr int RC = GEMV_RC;
  const int lane = threadIdx.x & 63;
  const int col0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RC;
  const int row = blockIdx.y;
  if (col0 >= n) return;
  const int64_t g = blockIdx.z;
  x += g * xs; w += g * ws;
  if (b) b += g * bs;
  if (y) y += g * ys;
  if (yh) yh += g * ys;
  const f16 *xr = x + (int64_t)row * ldx;
  const f16 *wr[RC];
#pragma unroll
  for (int c = 0; c < RC; ++c) wr[c] = w + (int64_t)min(col0 + c, n - 1) * k;   // (columns past n: recomputed, not stored)
  float acc[RC];
#pragma unroll
  for (int c = 0; c < RC; ++c) acc[c] = 0.f;
  for (int i = lane * 8; i < k; i += 64 * 8) {
    f16x8 wv[RC];
#pragma unroll
    for (int c = 0; c < RC; ++c) wv[c] = *(const f16x8 *)(wr[c] + i);
    const f16x8 xv = *(const f16x8 *)(xr + i);
    float xf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xf[e] = (float)xv[e];
      // torch evaluates F.silu on the fp16 tensor: round the activation back to fp16
      if (silu_in) xf[e] = (float)(f16)silu_f(xf[e]);
    }
#pragma unroll
    for (int c = 0; c < RC; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[c] += xf[e] * (float)wv[c][e];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int c = 0; c < RC; ++c) acc[c] += __shfl_xor(acc[c], o, 64);
#pragma unroll
  for (int c = 0; c < RC; ++c)
    if (lane == c && col0 + c < n) {
      float v = acc[c] + (b ? b[col0 + c] : 0.f);
      if (silu_out) v = silu_f(v);
      if (y) y[(int64_t)row * ldy + col0 + c] = v;
      if (yh) yh[(int64_t)row * ldy + col0 + c] = (f16)v;
    }
}

__global__ void sinusoid_kernel(const float *__restrict__ vals, f16 *__restrict__ out, int count, int dim) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (idx >= count * half) return;
  const int r = idx / half, j = idx - r * half;
  const float freq = expf(-9.210340371976184f * (float)j / (float)half);  // ln(10000)
  const float arg = vals[r] * freq;
  out[(int64_t)r * dim + j] = (f16)cosf(arg);
  out[(int64_t)r * dim + half + j] = (f16)sinf(arg);
}

}  // namespace

extern "C" int sp_pack_input_f16(const void *latent, const void *image_latents, void *out, float in_scale,
                                 int b, int frames, int h, int w, int cpad, void *stream) {
  SP_REQUIRE(latent && image_latents && out, "sp_pack_input_f16: null pointer");
  SP_REQUIRE(b > 0 && frames > 0 && h > 0 && w > 0, "sp_pack_input_f16: dims must be positive");
  SP_REQUIRE(cpad >= 8 && cpad % 8 == 0, "sp_pack_input_f16: cpad=%d must be a multiple of 8, >= 8", cpad);
  const int64_t hw = (int64_t)h * w, total = (int64_t)b * frames * hw;
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(pack_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const f16 *)latent, (const f16 *)image_latents, (f16 *)out,
                     in_scale, frames, hw, cpad, total);
  SP_CHECK_LAUNCH("sp_pack_input_f16");
  return SP_OK;
}

extern "C" int sp_euler_step_f16(const void *latent, const void *eps_cond, const void *eps_uncond,
                                 int64_t ld_eps, const float *guidance, void *out, float sigma,
                                 float sigma_next, int b, int frames, int h, int w, void *stream) {
  SP_REQUIRE(latent && eps_cond && out, "sp_euler_step_f16: null pointer");
  SP_REQUIRE(!eps_uncond || guidance, "sp_euler_step_f16: CFG needs a guidance vector");
  SP_REQUIRE(ld_eps >= 4 && ld_eps % 4 == 0, "sp_euler_step_f16: ld_eps must be a multiple of 4");
  SP_REQUIRE(sigma > 0.f, "sp_euler_step_f16: sigma must be positive");
  const int64_t hw = (int64_t)h * w, total = (int64_t)b * frames * hw;
  const float s2 = sigma * sigma + 1.0f;
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(euler_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const f16 *)latent, (const f16 *)eps_cond,
                     (const f16 *)eps_uncond, ld_eps, guidance, (f16 *)out, -sigma / sqrtf(s2), 1.0f / s2,
                     1.0f / sigma, sigma_next - sigma, frames, hw, total);
  SP_CHECK_LAUNCH("sp_euler_step_f16");
  return SP_OK;
}

extern "C" int sp_concat_channels_f16(const void *a, int ca, const void *b, int cb, void *out, int64_t rows,
                                      void *stream) {
  SP_REQUIRE(a && b && out, "sp_concat_channels_f16: null pointer");
  SP_REQUIRE(ca > 0 && cb > 0 && ca % 8 == 0 && cb % 8 == 0, "sp_concat_channels_f16: channels must be multiples of 8");
  const int64_t total = rows * ((ca + cb) / 8);
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(concat_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const f16 *)a, ca / 8, (const f16 *)b, cb / 8, (f16 *)out, total);
  SP_CHECK_LAUNCH("sp_concat_channels_f16");
  return SP_OK;
}

extern "C" int sp_add_rowvec_f16(const void *x, const float *vec, void *y, int64_t rows, int c, void *stream) {
  SP_REQUIRE(x && vec && y && rows > 0 && c % 8 == 0, "sp_add_rowvec_f16: bad arguments");
  const int64_t total = rows * (c / 8);
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(add_rowvec_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const f16 *)x, vec, (f16 *)y, c / 8, total);
  SP_CHECK_LAUNCH("sp_add_rowvec_f16");
  return SP_OK;
}

extern "C" int sp_gemv_f16(const void *x, int64_t ldx, const void *w, const float *b, float *y, void *y_f16,
                           int64_t ldy, int rows, int n, int k, int silu_in, int silu_out, void *stream) {
  SP_REQUIRE(x && w && (y || y_f16), "sp_gemv_f16: null pointer");
  SP_REQUIRE(rows > 0 && n > 0 && k > 0 && k % 8 == 0 && ldx % 8 == 0, "sp_gemv_f16: bad shape rows=%d n=%d k=%d", rows, n, k);
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(gemv_kernel, dim3((n + 4 * GEMV_RC - 1) / (4 * GEMV_RC), rows), dim3(256), 0, (hipStream_t)stream, (const f16 *)x,
                     ldx, (const f16 *)w, b, y, (f16 *)y_f16, ldy, n, k, silu_in, silu_out, (int64_t)0, (int64_t)0,
                     (int64_t)0, (int64_t)0);
  SP_CHECK_LAUNCH("sp_gemv_f16");
  return SP_OK;
}

extern "C" int sp_gemv_batched_f16(const void *x, int64_t ldx, int64_t x_stride, const void *w, int64_t w_stride,
                                   const float *b, int64_t b_stride, float *y, void *y_f16, int64_t ldy,
                                   int64_t y_stride, int batch, int rows, int n, int k, int silu_in, int silu_out,
                                   void *stream) {
  SP_REQUIRE(x && w && (y || y_f16), "sp_gemv_batched_f16: null pointer");
  SP_REQUIRE(batch > 0 && batch <= 65535 && rows > 0 && rows <= 65535 && n > 0 && k > 0 && k % 8 == 0 && ldx % 8 == 0 &&
                 x_stride % 8 == 0 && w_stride % 8 == 0,
             "sp_gemv_batched_f16: bad shape batch=%d rows=%d n=%d k=%d", batch, rows, n, k);
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(gemv_kernel, dim3((n + 4 * GEMV_RC - 1) / (4 * GEMV_RC), rows, batch), dim3(256), 0, (hipStream_t)stream, (const f16 *)x,
                     ldx, (const f16 *)w, b, y, (f16 *)y_f16, ldy, n, k, silu_in, silu_out, x_stride, w_stride,
                     b_stride, y_stride);
  SP_CHECK_LAUNCH("sp_gemv_batched_f16");
  return SP_OK;
}

extern "C" int sp_sinusoid_f16(const float *values, void *out, int count, int dim, void *stream) {
  SP_REQUIRE(values && out && count > 0 && dim > 0 && dim % 2 == 0, "sp_sinusoid_f16: bad arguments");
  const int total = count * (dim / 2);
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(sinusoid_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, values,
                     (f16 *)out, count, dim);
  SP_CHECK_LAUNCH("sp_sinusoid_f16");
  return SP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Measurement aid (bench.py `roofline.clock_ghz_live`): a time stamp of both counters a wave can read -- the shader-clock
// counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) -- from `blocks` one-wave workgroups, each with the
// id of the XCD it ran on (the shader-clock counters of different XCDs are not assumed to share an origin).  Two stamps on
// the same XCD, one before and one after a stretch of work, give the shader clock the chip held in between:
// 0.1 GHz x (shader ticks) / (100 MHz ticks).  No persistent kernel: a stamp is a ~2 us launch in stream order.
namespace {
__global__ void clock_stamp_kernel(unsigned long long *out) {
  if (threadIdx.x != 0) return;
  unsigned long long *o = out + (size_t)blockIdx.x * 4;
  o[0] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20);     // HW_REG_XCC_ID, bits 3:0
  o[1] = __builtin_amdgcn_s_memtime();
  o[2] = wall_clock64();
  o[3] = 1;
}
}  // namespace

extern "C" int sp_clock_stamp(void *out, int blocks, void *stream) {
  SP_REQUIRE(out && ((uintptr_t)out & 7) == 0, "sp_clock_stamp: out must be an 8-byte aligned device pointer");
  SP_REQUIRE(blocks > 0 && blocks <= 1024, "sp_clock_stamp: blocks=%d outside (0, 1024]", blocks);
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(clock_stamp_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, (unsigned long long *)out);
  SP_CHECK_LAUNCH("sp_clock_stamp");
  return SP_OK;
}

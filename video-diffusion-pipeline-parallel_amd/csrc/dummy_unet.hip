// DummyUNet forward (simulator-path model) in fp32 on gfx950, layout (B, C, F, H, W).
//   out = x + gain * Conv3d_2(SiLU(Conv3d_1(x))) + LayerNorm_C(x)
// Reference: /root/reference/src/models/dummy_unet.py:37-59.  Two launches:
//   conv1_silu : one thread per (b, f, y, x); all `hidden` output channels in registers (<= 64)
//   conv2_tail : one thread per (b, f, y, x); all C output channels, then the residual / LayerNorm
//                tail for that pixel (channel reduction in registers, no cross-lane traffic).
// Weights are tiny (C*hidden*27 floats) and are read through the scalar/L1 path.
#include "common.h"

namespace {

constexpr int MAXC = 64;

template <bool TAIL>
__global__ __launch_bounds__(256) void dummy_conv_kernel(
    const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
    float *__restrict__ out, const float *__restrict__ x0, const float *__restrict__ ln_w,
    const float *__restrict__ ln_b, float ln_eps, int use_ln, float gain, int B, int cin, int cout, int F,
    int H, int W) {
  const int64_t vol = (int64_t)F * H * W;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * vol) return;
  const int b = (int)(idx / vol);
  const int64_t p = idx - (int64_t)b * vol;
  const int f = (int)(p / ((int64_t)H * W));
  const int rem = (int)(p - (int64_t)f * H * W);
  const int y = rem / W, x = rem - y * W;
  float acc[MAXC];
#pragma unroll
  for (int co = 0; co < MAXC; ++co) acc[co] = co < cout ? bias[co] : 0.f;
  for (int ci = 0; ci < cin; ++ci) {
    const float *ip = in + ((int64_t)b * cin + ci) * vol;
    for (int kf = 0; kf < 3; ++kf) {
      const int ff = f + kf - 1;
      if (ff < 0 || ff >= F) continue;
      for (int kh = 0; kh < 3; ++kh) {
        const int yy = y + kh - 1;
        if (yy < 0 || yy >= H) continue;
        for (int kw = 0; kw < 3; ++kw) {
          const int xx = x + kw - 1;
          if (xx < 0 || xx >= W) continue;
          const float v = ip[((int64_t)ff * H + yy) * W + xx];
          const float *wp = w + ((int64_t)ci * 27 + (kf * 3 + kh) * 3 + kw);
#pragma unroll
          for (int co = 0; co < MAXC; ++co)
            if (co < cout) acc[co] += v * wp[(int64_t)co * cin * 27];
        }
      }
    }
  }
  if (!TAIL) {
#pragma unroll
    for (int co = 0; co < MAXC; ++co)
      if (co < cout) out[((int64_t)b * cout + co) * vol + p] = acc[co] / (1.0f + expf(-acc[co]));
  } else {
    float mean = 0.f, var = 0.f;
    const float *xp = x0 + (int64_t)b * cout * vol + p;
    if (use_ln) {
      for (int c = 0; c < cout; ++c) mean += xp[(int64_t)c * vol];
      mean /= (float)cout;
      for (int c = 0; c < cout; ++c) { const float d = xp[(int64_t)c * vol] - mean; var += d * d; }
      var /= (float)cout;
    }
    const float rstd = 1.0f / sqrtf(var + ln_eps);
#pragma unroll
    for (int co = 0; co < MAXC; ++co)
      if (co < cout) {
        const float xv = xp[(int64_t)co * vol];
        float v = xv + gain * acc[co];
        if (use_ln) v += (xv - mean) * rstd * ln_w[co] + ln_b[co];
        out[((int64_t)b * cout + co) * vol + p] = v;
      }
  }
}

}  // namespace

extern "C" int sp_dummy_unet_f32(const float *x, float *out, float *hidden, const float *w1, const float *b1,
                                 const float *w2, const float *b2, const float *ln_w, const float *ln_b,
                                 float ln_eps, int use_ln, float gain, int b, int c, int hidden_c, int frames,
                                 int h, int w, void *stream) {
  SP_REQUIRE(x && out && hidden && w1 && b1 && w2 && b2, "sp_dummy_unet_f32: null pointer");
  SP_REQUIRE(!use_ln || (ln_w && ln_b), "sp_dummy_unet_f32: LayerNorm parameters missing");
  SP_REQUIRE(b > 0 && frames > 0 && h > 0 && w > 0, "sp_dummy_unet_f32: dims must be positive");
  SP_REQUIRE(c > 0 && c <= MAXC && hidden_c > 0 && hidden_c <= MAXC,
             "sp_dummy_unet_f32: channels (%d) / hidden (%d) must be in [1,%d]", c, hidden_c, MAXC);
  const int64_t total = (int64_t)b * frames * h * w;
  const unsigned grid = (unsigned)((total + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(dummy_conv_kernel<false>, dim3(grid), dim3(256), 0, s, x, w1, b1, hidden, nullptr, nullptr,
                     nullptr, 0.f, 0, 0.f, b, c, hidden_c, frames, h, w);
  SP_CHECK_LAUNCH("sp_dummy_unet_f32(conv1)");
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(dummy_conv_kernel<true>, dim3(grid), dim3(256), 0, s, hidden, w2, b2, out, x, ln_w, ln_b,
                     ln_eps, use_ln, gain, b, hidden_c, c, frames, h, w);
  SP_CHECK_LAUNCH("sp_dummy_unet_f32(conv2)");
  return SP_OK;
}

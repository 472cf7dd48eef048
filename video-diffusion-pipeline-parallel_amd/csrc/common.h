// Shared helpers for the gfx950 kernels of libsvdpipe_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/svdpipe.h"

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WAVE 64

void sp_set_error(const char *fmt, ...);

#define SP_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      sp_set_error(__VA_ARGS__);              \
      return SP_EINVAL;                       \
    }                                         \
  } while (0)

// hipGetLastError() reports the last error of ANY earlier runtime call on this thread (PyTorch,
// profilers, ...); reading it here resets it so SP_CHECK_LAUNCH only sees this launch's status.
#define SP_CLEAR_STALE_ERROR() ((void)hipGetLastError())

#define SP_CHECK_LAUNCH(name)                                             \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess) {                                              \
      sp_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return SP_ELAUNCH;                                                  \
    }                                                                     \
  } while (0)

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, far below fp16 resolution): one rcp, one exp2,
// six fma – about a third of the instructions of ocml erff, which matters in the GEGLU GEMM epilogue.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  return copysignf(fmaf(-p, e, 1.0f), x);
}
// exact-form (erf) GELU, as torch F.gelu(approximate="none")
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erf_fast(v * 0.70710678118654752f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// global -> LDS direct load, 16 bytes per lane (global_load_lds_dwordx4).
// LDS destination = wave-uniform base + lane*16; the global source address is per lane.
__device__ __forceinline__ void glds16(const void *gsrc, void *lds_dst) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void *)gsrc,
      (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

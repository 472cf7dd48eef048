// Shared helpers for the gfx950 kernels of libsvdpipe_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/svdpipe.h"

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WAVE 64

void sp_set_error(const char *fmt, ...);

constexpr int SP_MAX_DEVICES = 64;
int sp_ensure_dyn_lds(const void *kernel, int bytes, bool (&done)[SP_MAX_DEVICES], const char *name);

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

// SiLU with v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 VALU ops): the GroupNorm apply pass runs it on
// every element and is otherwise HBM-bound; the result is rounded to fp16 anyway.
__device__ __forceinline__ float silu_f(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}
// exact-form (erf) GELU, as torch F.gelu(approximate="none"), written for the GEGLU GEMM epilogue where it is the
// dominant VALU cost: gelu(g) = g*Phi(g) = max(g,0) - |g|*(1 - Phi(|g|)) = g/2 + |g|*(1/2 - (1 - Phi(|g|))) and
// 1 - Phi(t) = exp2(q(t)) with q a
// degree-6 fit of log2(erfc(t/sqrt2)/2) on [0, 5.6] (beyond that the term is < 1e-8 and t is clamped).
// |abs error| <= 5e-7 over all g (checked against scipy in float32), one transcendental and ten VALU ops.
__device__ __forceinline__ float gelu_f(float v) {
  // (v_med3_f32 instead of fminf/fmaxf: the libm forms add a canonicalising v_max_f32 x,x each)
  const float a = __builtin_amdgcn_fmed3f(fabsf(v), 0.0f, 5.6f);
  float q = 3.470272457e-05f;
  q = fmaf(q, a, -7.831060430e-04f);
  q = fmaf(q, a, 8.125715224e-03f);
  q = fmaf(q, a, -5.348086292e-02f);
  q = fmaf(q, a, -4.587201634e-01f);
  q = fmaf(q, a, -1.151218199e+00f);
  q = fmaf(q, a, -9.999913501e-01f);
  // max(v,0) - |v|*e  ==  0.5*v + |v|*(0.5 - e): three VALU ops, no compare/select and no canonicalising max
  return fmaf(fabsf(v), 0.5f - __builtin_amdgcn_exp2f(q), 0.5f * v);
}

// Two GELUs at once on packed-fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of fp32 per VALU issue);
// same polynomial, same rounding as gelu_f.  The GEGLU epilogues are VALU-bound, the fma chain is 60 % of their work.
__device__ __forceinline__ f32x2 gelu2_f(f32x2 v) {
  const f32x2 av = {fabsf(v[0]), fabsf(v[1])};
  const f32x2 a = {__builtin_amdgcn_fmed3f(av[0], 0.0f, 5.6f), __builtin_amdgcn_fmed3f(av[1], 0.0f, 5.6f)};
  f32x2 q = {3.470272457e-05f, 3.470272457e-05f};
  q = __builtin_elementwise_fma(q, a, (f32x2){-7.831060430e-04f, -7.831060430e-04f});
  q = __builtin_elementwise_fma(q, a, (f32x2){8.125715224e-03f, 8.125715224e-03f});
  q = __builtin_elementwise_fma(q, a, (f32x2){-5.348086292e-02f, -5.348086292e-02f});
  q = __builtin_elementwise_fma(q, a, (f32x2){-4.587201634e-01f, -4.587201634e-01f});
  q = __builtin_elementwise_fma(q, a, (f32x2){-1.151218199e+00f, -1.151218199e+00f});
  q = __builtin_elementwise_fma(q, a, (f32x2){-9.999913501e-01f, -9.999913501e-01f});
  const f32x2 e = {0.5f - __builtin_amdgcn_exp2f(q[0]), 0.5f - __builtin_amdgcn_exp2f(q[1])};
  return __builtin_elementwise_fma(av, e, v * 0.5f);
}

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

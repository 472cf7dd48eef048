// Does one wave overlap its own MFMAs with its VALU / transcendental instructions on gfx950, and what do v_exp_f32 and
// v_cvt_pk_f16_f32 cost next to a 32x32x16 MFMA?  Straight-line loops, one to three waves per SIMD, wall-clock per
// iteration.   hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SB() __builtin_amdgcn_sched_barrier(0)

// MODE 0: 8 MFMA | 1: 32 exp | 2: 16 cvt_pk | 3: 8 x (MFMA, 4 exp) | 4: 8 MFMA then 32 exp | 5: 8 x (MFMA, 4 exp, 2 cvt)
// 6: 8 MFMA then 32 exp then 16 cvt | 7: 32 v_max_f32 | 8: 8 x (MFMA, 4 v_max) | 9: 8 x (MFMA, 2 exp, 1 cvt)
// 10 / 11 / 12: 8 MFMA over 1 / 4 / 8 accumulators (modes 0-9 alternate two: every MFMA depends on the one before last)
template <int MODE>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float seed) {
  f32x16 acc[8];
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(seed + threadIdx.x * 1e-3f); b[e] = (_Float16)(seed * 0.5f); }
  for (int a = 0; a < 8; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = seed * (a + 1);
  float x[32];
  for (int e = 0; e < 32; ++e) x[e] = seed * (e + 1) * 1e-3f;
  unsigned pk[16];
  for (int e = 0; e < 16; ++e) pk[e] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 0 || MODE == 3 || MODE == 4 || MODE == 5 || MODE == 6 || MODE == 8 || MODE == 9)
        acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j & 1], 0, 0, 0);
      if (MODE >= 10) {
        constexpr int NA = MODE == 10 ? 1 : (MODE == 11 ? 4 : 8);
        acc[j % NA] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j % NA], 0, 0, 0);
      }
      if (MODE == 3 || MODE == 5) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[4 * j + e] = __builtin_amdgcn_exp2f(x[4 * j + e]);
      }
      if (MODE == 9) {
#pragma unroll
        for (int e = 0; e < 2; ++e) x[2 * j + e] = __builtin_amdgcn_exp2f(x[2 * j + e]);
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        f32x2 p = {x[2 * j], x[2 * j + 1]};
        pk[j] ^= __builtin_bit_cast(unsigned, __builtin_convertvector(p, f16x2));
      }
      if (MODE == 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[4 * j + e] = fmaxf(x[4 * j + e], x[(4 * j + e + 1) & 31]);
      }
      if (MODE == 5) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          f32x2 p = {x[4 * j + 2 * e], x[4 * j + 2 * e + 1]};
          pk[2 * j + e] ^= __builtin_bit_cast(unsigned, __builtin_convertvector(p, f16x2));
        }
      }
      SB();
    }
    if (MODE == 1 || MODE == 4 || MODE == 6) {
#pragma unroll
      for (int e = 0; e < 32; ++e) x[e] = __builtin_amdgcn_exp2f(x[e]);
      SB();
    }
    if (MODE == 7) {
#pragma unroll
      for (int e = 0; e < 32; ++e) x[e] = fmaxf(x[e], x[(e + 1) & 31]);
      SB();
    }
    if (MODE == 2 || MODE == 6) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        f32x2 p = {x[2 * e], x[2 * e + 1]};
        pk[e] ^= __builtin_bit_cast(unsigned, __builtin_convertvector(p, f16x2));
      }
      SB();
    }
  }
  float s = 0.f;
  for (int a = 0; a < 8; ++a)
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  for (int e = 0; e < 16; ++e) s += __uint_as_float(pk[e]);
  for (int e = 0; e < 32; ++e) s += x[e];
  if (s == 12345.678f) out[threadIdx.x] = s;       // never true: keeps the work alive
}

template <int MODE>
double run(int wgs_per_cu, int iters, float *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 0.37f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 0.37f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / iters;                          // ns per iteration (all resident waves run concurrently)
}

int main() {
  float *out; hipMalloc(&out, 4096);
  const int iters = 20000;
  const char *names[] = {"8 MFMA 32x32x16", "32 v_exp_f32", "16 v_cvt_pk_f16_f32", "8 x (MFMA, 4 exp)", "8 MFMA; 32 exp",
                         "8 x (MFMA, 4 exp, 2 cvt)", "8 MFMA; 32 exp; 16 cvt", "32 v_max_f32", "8 x (MFMA, 4 v_max)",
                         "8 x (MFMA, 2 exp, 1 cvt)", "8 MFMA, one accumulator", "8 MFMA, four accumulators",
                         "8 MFMA, eight accumulators"};
  for (int w = 1; w <= 3; ++w) {
    printf("---- %d wave(s) per SIMD (grid = %d workgroups of 256)\n", w, 256 * w);
    double t[13] = {run<0>(w, iters, out), run<1>(w, iters, out), run<2>(w, iters, out), run<3>(w, iters, out), run<4>(w, iters, out),
                    run<5>(w, iters, out), run<6>(w, iters, out), run<7>(w, iters, out), run<8>(w, iters, out), run<9>(w, iters, out),
                    run<10>(w, iters, out), run<11>(w, iters, out), run<12>(w, iters, out)};
    for (int m = 0; m < 13; ++m) printf("  %-28s %8.1f ns / iteration\n", names[m], t[m]);
  }
  return 0;
}

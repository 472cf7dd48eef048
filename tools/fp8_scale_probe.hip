// Would spatial attention on the block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, the only fp8 form that runs at
// 2x the fp16 rate on gfx950 -- MI355X_MICROARCH.md, matrix-core table) pay at head_dim 64?  Two questions, one binary:
//
//  (1) operand layout: with unit scales (E8M0 127), lane l supplies row (A) / column (B) l & 31 and 32 of the 64 k values in
//      its 8 operand registers, byte j of lane half h of A meeting byte j of lane half h of B.  Checked exactly with
//      small-integer e4m3 data against a host product, under two assignments of k to (h, j): both are exact, as any
//      assignment must be that is applied to both operands alike (the instruction sums over k) -- which is all an
//      attention kernel needs: P's keys come out of the score accumulators in a fixed (h, j) order and V is stored to match.
//  (2) price of one (64 keys x 32 queries) tile of the attention inner loop per wave, 1 to 3 waves per SIMD:
//        f16     : 16 x v_mfma_f32_32x32x16_f16     + 32 v_exp_f32 + 16 v_cvt_pk_f16_f32 + 16 v_max3_f32
//        fp8     : 16 x v_mfma_f32_32x32x16_fp8_fp8 + 32 v_exp_f32 + 16 v_cvt_pk_fp8_f32 + 16 v_max3_f32   (today's kernel)
//        fp8 x64 :  4 x v_mfma_scale_f32_32x32x64   + 32 v_exp_f32 + 16 v_cvt_pk_fp8_f32 + 16 v_max3_f32   (the candidate)
//        valu    : the vector work alone
//      The vector instructions are spread evenly between the MFMAs (sched_barrier between steps).  LDS reads, LDS-DMA and
//      the barrier of the real loop are NOT in here: the ratio fp8 x64 / f16 is an upper bound on what the kernel can gain.
//
//   hipcc --offload-arch=gfx950 -O3 tools/fp8_scale_probe.hip -o /tmp/fp8_scale_probe && /tmp/fp8_scale_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

#define SB() __builtin_amdgcn_sched_barrier(0)

// ---------------------------------------------------------------------------------------------- (1) layout
// a8[m][k], b8[n][k]: e4m3 bytes, 32 x 64 each; c[m][n] = sum_k a*b
__global__ void layout_kernel(const unsigned char *a8, const unsigned char *b8, float *c, int hyp) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  i32x8 a, b;
  unsigned char *pa = (unsigned char *)&a, *pb = (unsigned char *)&b;
  for (int j = 0; j < 32; ++j) {
    // hypothesis 0: k = 32*h + j ; hypothesis 1: k = 16*h + (j & 15) + 32*(j >> 4)
    const int k = hyp == 0 ? 32 * h + j : 16 * h + (j & 15) + 32 * (j >> 4);
    pa[j] = a8[r * 64 + k];
    pb[j] = b8[r * 64 + k];
  }
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  // cbsz / blgp 0 = e4m3 on both operands; scales: E8M0 127 = 2^0
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 127, 0, 127);
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;      // C/D layout of every 32x32 form
    c[row * 32 + r] = acc[e];
  }
}

static float e4m3_value(unsigned char v) {
  static const float tab[5] = {0.f, 1.f, 2.f, -1.f, -2.f};
  static const unsigned char enc[5] = {0x00, 0x38, 0x40, 0xB8, 0xC0};
  for (int i = 0; i < 5; ++i)
    if (enc[i] == v) return tab[i];
  return 0.f;
}

static bool check_layout() {
  static const unsigned char enc[5] = {0x00, 0x38, 0x40, 0xB8, 0xC0};
  std::vector<unsigned char> a(32 * 64), b(32 * 64);
  srand(7);
  for (auto &v : a) v = enc[rand() % 5];
  for (auto &v : b) v = enc[rand() % 5];
  std::vector<float> want(32 * 32, 0.f), got(32 * 32);
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n)
      for (int k = 0; k < 64; ++k) want[m * 32 + n] += e4m3_value(a[m * 64 + k]) * e4m3_value(b[n * 64 + k]);
  unsigned char *da, *db;
  float *dc;
  hipMalloc(&da, a.size()); hipMalloc(&db, b.size()); hipMalloc(&dc, got.size() * 4);
  hipMemcpy(da, a.data(), a.size(), hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), b.size(), hipMemcpyHostToDevice);
  bool any = false;
  for (int hyp = 0; hyp < 2; ++hyp) {
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, da, db, dc, hyp);
    hipMemcpy(got.data(), dc, got.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32 * 32; ++i) bad += got[i] != want[i];
    printf("layout hypothesis %d (%s): %d of 1024 outputs differ from the host product%s\n", hyp,
           hyp == 0 ? "lane half h holds k = 32h .. 32h+31" : "lane half h holds k = 16h..16h+15 and 32+16h..",
           bad, bad == 0 ? "  <-- this is the operand layout" : "");
    any |= bad == 0;
  }
  return any;
}

// ---------------------------------------------------------------------------------------------- (2) prices
// KIND 0 f16, 1 fp8 32x32x16, 2 fp8 scaled 32x32x64, 3 vector work alone
template <int KIND>
__global__ __launch_bounds__(256) void price_kernel(float *out, int iters, float seed) {
  f32x16 acc[2];
  for (int a = 0; a < 2; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = seed * (a + 1);
  f16x8 ha, hb;
  for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)(seed + threadIdx.x * 1e-3f); hb[e] = (_Float16)(seed * 0.5f); }
  long qa = 0x3839404038393840L + threadIdx.x, qb = 0x4038383940383839L;
  i32x8 wa, wb;
  for (int e = 0; e < 8; ++e) { wa[e] = 0x38394040 + threadIdx.x + e; wb[e] = 0x40383839 + e; }
  float x[32], mx = seed;
  for (int e = 0; e < 32; ++e) x[e] = seed * (e + 1) * 1e-3f;
  unsigned pk[16];
  for (int e = 0; e < 16; ++e) pk[e] = 0;
  constexpr int NM = KIND == 2 ? 4 : 16;                 // MFMAs per tile; vector work per MFMA step = 1/NM of the tile's
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      if (KIND == 0) acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[j & 1], 0, 0, 0);
      if (KIND == 1) acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(qa, qb, acc[j & 1], 0, 0, 0);
      if (KIND == 2) acc[j & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa, wb, acc[j & 1], 0, 0, 0, 127, 0, 127);
      constexpr int EX = 32 / NM, CV = 16 / NM, MX = 16 / NM;
#pragma unroll
      for (int e = 0; e < MX; ++e) mx = __builtin_fmaxf(__builtin_fmaxf(mx, x[(EX * j + 2 * e) & 31]), x[(EX * j + 2 * e + 1) & 31]);
#pragma unroll
      for (int e = 0; e < EX; ++e) x[EX * j + e] = __builtin_amdgcn_exp2f(x[EX * j + e] - mx * 1e-9f);
#pragma unroll
      for (int e = 0; e < CV; ++e) {
        const int i = CV * j + e;
        if (KIND == 0 || KIND == 3) {
          f32x2 p = {x[2 * i], x[2 * i + 1]};
          pk[i] ^= __builtin_bit_cast(unsigned, __builtin_convertvector(p, f16x2));
        } else {
          if (i & 1) pk[i & ~1] = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], pk[i & ~1], true);
          else pk[i & ~1] = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], pk[i & ~1], false);
        }
      }
      SB();
    }
  }
  float s = mx;
  for (int a = 0; a < 2; ++a)
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  for (int e = 0; e < 16; ++e) s += __uint_as_float(pk[e]);
  for (int e = 0; e < 32; ++e) s += x[e];
  if (s == 12345.678f) out[threadIdx.x] = s;             // never true: keeps the work alive
}

template <int KIND>
double price(int waves_per_simd, int iters, float *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * waves_per_simd;
  hipLaunchKernelGGL(price_kernel<KIND>, dim3(grid), dim3(256), 0, 0, out, iters, 0.37f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(price_kernel<KIND>, dim3(grid), dim3(256), 0, 0, out, iters, 0.37f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / iters;                               // ns per tile per wave (all resident waves run concurrently)
}

int main() {
  const bool ok = check_layout();
  float *out;
  hipMalloc(&out, 4096);
  const int iters = 20000;
  const char *names[4] = {"f16     (16 MFMA 32x32x16)", "fp8     (16 MFMA 32x32x16)", "fp8 x64 ( 4 MFMA 32x32x64 scaled)", "vector work alone"};
  for (int w = 1; w <= 3; ++w) {
    const double t[4] = {price<0>(w, iters, out), price<1>(w, iters, out), price<2>(w, iters, out), price<3>(w, iters, out)};
    printf("---- %d wave(s) per SIMD: ns per (64 keys x 32 queries) tile and wave; tiles per SIMD and us\n", w);
    for (int k = 0; k < 4; ++k) printf("  %-36s %8.1f ns   %6.2f tiles/us/SIMD   x%.2f vs f16\n", names[k], t[k], w * 1e3 / t[k], t[0] / t[k]);
  }
  return ok ? 0 : 1;
}

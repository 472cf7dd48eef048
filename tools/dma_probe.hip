// Probe: LDS-DMA (global_load_lds_dwordx4) throughput per CU for two source shapes at equal bytes:
//   mode 0: 1 KiB piece = 8 rows x 128 B (full cache lines)      mode 1: 1 KiB piece = 16 rows x 64 B (half lines)
// Each workgroup (512 threads) streams `rows` rows of a [R][K] fp16 matrix like a GEMM A/B tile would.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ void glds16(const void *g, void *l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                   (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}
template <int MODE>
__global__ __launch_bounds__(512) void probe(const char *base, long row_bytes, int rows_total, int ksteps, int share) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // block b reads the row panel (b / share): `share` blocks read the same panel (L2 reuse like GEMM tiles)
  const int panel = blockIdx.x / share;
  const long row0 = ((long)panel * 512) % rows_total;
  for (int ks = 0; ks < ksteps; ++ks) {
    char *slot = smem + (ks & 3) * 32768;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char *src;
      if (MODE == 0) {  // 8 rows x 128 B; this K-step covers 64 B per row for 512 rows -> here 256 rows x 128 B
        const int r = (i * 8 + wave) * 8 + (lane >> 3);
        src = base + (row0 + r) * row_bytes + (long)ks * 128 + (lane & 7) * 16;
      } else if (MODE == 1) {          // 16 rows x 64 B, 512 rows x 64 B per K-step
        const int r = (i * 8 + wave) * 16 + (lane >> 2);
        src = base + (row0 + r) * row_bytes + (long)ks * 64 + (lane & 3) * 16;
      } else {          // MODE 2: like mode 1 but two consecutive K-steps (same lines) issued back to back
        const int r = (i * 8 + wave) * 16 + (lane >> 2);
        src = base + (row0 + r) * row_bytes + (long)(ks * 2) * 64 + (lane & 3) * 16;
        glds16(src, slot + (i * 8 + wave) * 1024);
        src += 0;  // first half issued; the second half of the same lines follows in the second loop below
      }
      if (MODE != 2) glds16(src, slot + (i * 8 + wave) * 1024);
    }
    if (MODE == 2) {
      char *slot2 = smem + ((ks + 2) & 3) * 32768;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = (i * 8 + wave) * 16 + (lane >> 2);
        glds16(base + (row0 + r) * row_bytes + (long)(ks * 2 + 1) * 64 + (lane & 3) * 16, slot2 + (i * 8 + wave) * 1024);
      }
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && smem[0] == 123) printf("x");
}
int main(int argc, char **argv) {
  const int share = argc > 1 ? atoi(argv[1]) : 8;
  const long K = 4096, row_bytes = K * 2;
  const int rows_total = 65536;
  char *d; hipMalloc(&d, rows_total * row_bytes); hipMemset(d, 1, rows_total * row_bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    const int ksteps = mode == 1 ? 128 : 64;      // both walk 8 KB per row
    const int blocks = 256 * 4;
    auto launch = [&]() {
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(512), 131072, 0, d, row_bytes, rows_total, ksteps, share);
      else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(512), 131072, 0, d, row_bytes, rows_total, ksteps, share);
      else hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(512), 131072, 0, d, row_bytes, rows_total, ksteps, share);
    };
    hipFuncSetAttribute((const void *)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double bytes = (double)blocks * ksteps * 32768.0 * (mode == 2 ? 2 : 1);
    printf("mode %d (%s) share %d: %.1f us, %.2f TB/s aggregate, %.1f GB/s per CU\n", mode,
           mode == 0 ? "8 rows x 128 B" : mode == 1 ? "16 rows x 64 B" : "16x64B, k and k+1 paired", share, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
  }
  return 0;
}

// Probe: how fast can a 512-thread workgroup per CU write a 256 x 256 fp16 output tile of a [M][ld] matrix, by the
// shape of its store instructions (all dwordx4 unless noted; same bytes, same addresses overall):
//   A  full rows      : thread i of 512 writes 16 B chunk (i % 32) of row (i / 32) + 16*it  (what the LDS-staged epilogue does)
//   B  transposed 4x4 : wave (wm, wn), lane (fr, fq), j = 0..7: 2 stores of 16 B at row wm*128 + j*16 + fr, columns
//                       wn*64 + fq*16 + {0, 8}          (each instruction: 16 rows x 4 pieces of 16 B at 32 B stride)
//   C  8-byte direct  : j, i: row as B, columns wn*64 + i*16 + 4*fq, dwordx2   (the MFMA accumulator layout as it stands)
//   D  transposed, 64-B runs: as B but the two stores cover columns wn*64 + 8*fq and wn*64 + 32 + 8*fq
//                       (each instruction: 16 rows x 64 contiguous bytes)
// The kernel only stores (no loads, no LDS); time per tile = how long the chip needs to absorb one round of tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(_Float16 *d, long ld, int tiles_n, int reps) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fq = lane >> 4;
  for (int rep = 0; rep < reps; ++rep) {
    const int t = blockIdx.x + rep * gridDim.x;
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    _Float16 *base = d + (long)tm * 256 * ld + tn * 256;
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (_Float16)(tid + e + rep);
    if (MODE == 0) {
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int idx = tid + it * 512, r = idx >> 5, c = idx & 31;
        *(f16x8 *)(base + (long)r * ld + c * 8) = v;
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        _Float16 *p = base + (long)(wm * 128 + j * 16 + fr) * ld + wn * 64 + fq * 16;
        *(f16x8 *)p = v;
        *(f16x8 *)(p + 8) = v;
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *(f16x4 *)(base + (long)(wm * 128 + j * 16 + fr) * ld + wn * 64 + i * 16 + 4 * fq) = (f16x4){v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        _Float16 *p = base + (long)(wm * 128 + j * 16 + fr) * ld + wn * 64 + fq * 8;
        *(f16x8 *)p = v;
        *(f16x8 *)(p + 32) = v;
      }
    }
  }
}

int main(int argc, char **argv) {
  const long M = 129024;
  const long lds[3] = {2560, 1280, 320};
  _Float16 *d; hipMalloc(&d, M * 2560 * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[4] = {"A full rows", "B transposed 4x4 (16 B pieces)", "C 8-byte direct", "D transposed, 64-B runs"};
  for (int li = 0; li < 3; ++li) {
    const long ld = lds[li];
    const int tiles_n = ld >= 256 ? (int)(ld / 256) : 1, tiles = (int)(M / 256) * tiles_n;
    for (int persistent = 0; persistent < 2; ++persistent) {
      const int grid = persistent ? 256 : tiles, reps = persistent ? tiles / 256 : 1;
      for (int mode = 0; mode < 4; ++mode) {
        if (ld < 256) continue;
        auto launch = [&]() {
          switch (mode) {
            case 0: hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 0, 0, d, ld, tiles_n, reps); break;
            case 1: hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(512), 0, 0, d, ld, tiles_n, reps); break;
            case 2: hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(512), 0, 0, d, ld, tiles_n, reps); break;
            default: hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(512), 0, 0, d, ld, tiles_n, reps); break;
          }
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double bytes = (double)grid * reps * 256 * 256 * 2;
        printf("ld %4ld %-10s %-32s: %8.1f us  %.2f TB/s  %.2f us per round of 256 tiles\n", ld,
               persistent ? "persistent" : "grid=tiles", names[mode], ms * 1e3, bytes / ms / 1e9,
               ms * 1e3 / ((double)grid * reps / 256));
      }
    }
  }
  return 0;
}

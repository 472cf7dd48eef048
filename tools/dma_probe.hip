// Probe: how fast can one 512-thread workgroup per CU pull operand slabs through the vector memory path, by the SHAPE of
// the access (the question behind the GEMM K-step: is 43 GB/s per CU a limit of the path or of the 64-byte rows?)
//   shape 0: 16 rows x  64 B per wave instruction (what a BK = 32 fp16 K-step reads: half a 128-byte line per row)
//   shape 1:  8 rows x 128 B per wave instruction (full lines: a BK = 64 K-step)
//   shape 2:  4 rows x 256 B
//   shape 3:  1 KiB contiguous
//   shape 4: as shape 0, but every wave issues the two halves of a line back to back (the pieces of K-steps 2i and
//            2i+1 interleaved, two K-steps staged at a time): does the second half then hit in the CU's L1?
// x path 0 = global_load_lds_dwordx4 (LDS-DMA, 4-stage ring of 32 KiB), 1 = global_load_dwordx4 into registers
// x source 0 = L2-resident (every XCD's workgroups walk the same 2 MiB), 1 = HBM stream (fresh bytes per workgroup)
// x grid 256 (every CU) / 32 / 8 / 1 workgroups.
// Row stride 2560 B (K = 1280 fp16).  Every step a workgroup reads 32 KiB = 32 instructions of 1 KiB; consecutive
// steps advance along the row by the row width (the K direction), wrapping to the next row block after the row ends --
// so with 64-byte rows the second half of every line is asked for one step later, as in the GEMM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void *gsrc, void *lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

template <int SHAPE, int PATH>
__global__ __launch_bounds__(512) void probe(const char *src, int hbm, int steps, float *sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWB = (SHAPE == 0 || SHAPE == 4) ? 64 : SHAPE == 1 ? 128 : SHAPE == 2 ? 256 : 1024;   // bytes per row per instruction
  constexpr int RPI = 1024 / ROWB;                                                     // rows per instruction
  constexpr long LD = 2560;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane / (ROWB / 16), lchunk = lane % (ROWB / 16);
  // 32 instructions per step = 32 * RPI rows of ROWB bytes; a row block = 32 * RPI rows x LD bytes
  const long rows_per_step = 32 * RPI;
  const int steps_per_row = SHAPE == 3 ? 1 : (int)(LD / ROWB);
  // L2-resident: the workgroups of an XCD (blockIdx % 8) walk the same 2 MiB window, 64 KiB apart; HBM: 8 MiB each
  const char *base = hbm ? src + (long)blockIdx.x * (8L << 20) : src + (long)(blockIdx.x & 7) * (2L << 20);
  const long start = hbm ? 0 : (long)(blockIdx.x >> 3) * 65536;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  auto addr = [&](int step, int j) -> const char * {
    long off;
    if (SHAPE == 3) {
      off = (long)step * 32768 + (long)(j * 8 + wave) * 1024 + lane * 16;
    } else {
      const int blk = step / steps_per_row, kk = step - blk * steps_per_row;
      const long row = (long)blk * rows_per_step + (long)(j * 8 + wave) * RPI + lrow;
      off = row * LD + (long)kk * ROWB + lchunk * 16;
    }
    off += start;
    if (!hbm) off &= (2L << 20) - 1;
    return base + off;
  };
  if (PATH == 0 && SHAPE == 4) {
    // 4-stage ring, K-steps staged in pairs: (s, s+1) issued together, piece by piece
    auto pair = [&](int s) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        glds16(addr(s, j), smem + (s & 3) * 32768 + (j * 8 + wave) * 1024);
        glds16(addr(s + 1, j), smem + ((s + 1) & 3) * 32768 + (j * 8 + wave) * 1024);
      }
    };
    pair(0);
    for (int s = 0; s < steps; s += 2) {
      if (s + 2 < steps) {
        pair(s + 2);
        __builtin_amdgcn_s_waitcnt(0x0f70 | 8);         // vmcnt(8): steps s, s+1 have landed
      } else {
        __builtin_amdgcn_s_waitcnt(0x0f70);
      }
      __builtin_amdgcn_s_barrier();
      acc += *(const f32x4 *)(smem + (s & 3) * 32768 + tid * 16);
      acc += *(const f32x4 *)(smem + ((s + 1) & 3) * 32768 + tid * 16);
      __builtin_amdgcn_s_barrier();
    }
  } else if (PATH == 0) {
    // 4-stage ring, 3 steps in flight
    for (int s = 0; s < 3 && s < steps; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) glds16(addr(s, j), smem + (s & 3) * 32768 + (j * 8 + wave) * 1024);
    for (int s = 0; s < steps; ++s) {
      if (s + 3 < steps) {
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(addr(s + 3, j), smem + ((s + 3) & 3) * 32768 + (j * 8 + wave) * 1024);
        __builtin_amdgcn_s_waitcnt(0x0f70 | 12);        // vmcnt(12): step s has landed
      } else {
        __builtin_amdgcn_s_waitcnt(0x0f70);             // vmcnt(0)
      }
      __builtin_amdgcn_s_barrier();
      // touch the stage so the reads cannot be dropped (one ds_read per lane)
      acc += *(const f32x4 *)(smem + (s & 3) * 32768 + tid * 16);
      __builtin_amdgcn_s_barrier();
    }
  } else {
    for (int s = 0; s < steps; s += 2) {
      f32x4 v[8];
      if (SHAPE == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = *(const f32x4 *)addr(s, j); v[2 * j + 1] = *(const f32x4 *)addr(s + 1, j); }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *(const f32x4 *)addr(s, j);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 + j] = *(const f32x4 *)addr(s + 1 < steps ? s + 1 : s, j);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[blockIdx.x] = acc[0];
}

template <int SHAPE, int PATH>
float run(const char *src, int hbm, int steps, float *sink, int grid) {
  hipFuncSetAttribute((const void *)probe<SHAPE, PATH>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<SHAPE, PATH>), dim3(grid), dim3(512), 131072, 0, src, hbm, steps, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((probe<SHAPE, PATH>), dim3(grid), dim3(512), 131072, 0, src, hbm, steps, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / 3;
}

// `dma_probe calib`: the FETCH_SIZE calibration of profiles/r03_fetch_size_calibration.txt -- only the HBM-stream cases
// on all 256 CUs (every byte is read exactly once from memory: 256 x 240 x 32 KiB = 2,013,265,920 B per dispatch), so
// that `rocprofv3 --pmc FETCH_SIZE -- tools/dma_probe calib` gives counter-per-known-byte for each access shape.
int main(int argc, char **argv) {
  const bool calib = argc > 1 && argv[1][0] == 'c';
  const long total = 3L << 30;                     // 3 GiB source
  char *src; float *sink;
  if (hipMalloc(&src, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(src, 1, total);
  hipMalloc(&sink, 4096);
  const char *shapes[5] = {"16 rows x  64 B", " 8 rows x 128 B", " 4 rows x 256 B", "1 KiB contiguous", "16 x 64 B, halves paired"};
  const int grids[2] = {256, 32};
  for (int hbm = calib ? 1 : 0; hbm < 2; ++hbm) {
    const int steps = hbm ? 240 : 2048;
    for (int gi = 0; gi < (calib ? 1 : 2); ++gi) {
      const int grid = grids[gi];
      for (int path = 0; path < 2; ++path)
        for (int shape = 0; shape < 5; ++shape) {
          float ms;
          switch (shape * 2 + path) {
            case 0: ms = run<0, 0>(src, hbm, steps, sink, grid); break;
            case 1: ms = run<0, 1>(src, hbm, steps, sink, grid); break;
            case 2: ms = run<1, 0>(src, hbm, steps, sink, grid); break;
            case 3: ms = run<1, 1>(src, hbm, steps, sink, grid); break;
            case 4: ms = run<2, 0>(src, hbm, steps, sink, grid); break;
            case 5: ms = run<2, 1>(src, hbm, steps, sink, grid); break;
            case 6: ms = run<3, 0>(src, hbm, steps, sink, grid); break;
            case 7: ms = run<3, 1>(src, hbm, steps, sink, grid); break;
            case 8: ms = run<4, 0>(src, hbm, steps, sink, grid); break;
            default: ms = run<4, 1>(src, hbm, steps, sink, grid); break;
          }
          const double bytes = (double)grid * steps * 32768;
          printf("%-12s grid %3d  %-10s %s : %8.1f us  %7.2f TB/s  %6.1f GB/s per CU\n", hbm ? "HBM stream" : "L2-resident",
                 grid, path == 0 ? "LDS-DMA" : "registers", shapes[shape], ms * 1e3, bytes / ms / 1e9,
                 bytes / ms / 1e6 / grid);
          fflush(stdout);
        }
    }
  }
  return 0;
}

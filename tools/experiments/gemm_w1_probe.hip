// Timing probe for DESIGN section 8 item 1 / VERDICT r04 item 3(b): a persistent 256 x 256 linear contraction with ONE WAVE
// PER SIMD -- four waves of 128 x 128 outputs (256 accumulator registers each) instead of eight of 128 x 64 -- and a
// hand-placed issue stream: the LDS reads of K-step g+1 and the LDS-DMA pieces of K-step g+3 are spread between the 64 MFMAs
// of K-step g, one barrier per 32-deep K-step, the DMA stream runs on across tile boundaries.
//   per K-step and CU: 64 KB of ds_read instead of 96 (each wave reads 8 + 8 fragments for 64 MFMAs instead of 8 + 4 for 32)
//   the previous tile's packed outputs can stay in registers, so GEGLU + stores can run under the next tile's MFMAs (MODE 2)
// Results are CHECKED against a host reference on a sample of outputs (so the stream is a real contraction, not a stale-operand
// loop), then timed on random data.
//   usage: gemm_w1_probe [M N K] [reps]     (defaults 258048 2560 320: the level-0 FF1 of two videos)
// Modes timed: 0 = K loop + plain fp16 stores after each tile; 1 = K loop only (no stores: upper bound); 2 = GEGLU (value,
// gate interleaved by 16 columns) + stores of the previous tile spread under the next tile's K loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <type_traits>

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((__vector_size__(4 * sizeof(unsigned))));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void *gsrc, void *lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ int swz4(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
__device__ __forceinline__ unsigned pack_h2(float a, float b) { const f16x2 h = {(f16)a, (f16)b}; return __builtin_bit_cast(unsigned, h); }
__device__ __forceinline__ void swap16(unsigned &a, unsigned &b) {
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0]; b = r[1];
}
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

constexpr int BM = 256, BN = 256, SLOT = 32768, NSLOT = 4, DIST = 3;
#define SB() __builtin_amdgcn_sched_barrier(0)

// MODE 0: plain stores after the tile; 1: no stores; 2: GEGLU, previous tile's outputs stored under the next tile's K loop
template <int MODE, bool BDIRECT>
__global__ __launch_bounds__(256, 1) void w1_kernel(const f16 *__restrict__ A, const f16 *__restrict__ W, f16 *__restrict__ D,
                                                    int M, int N, int K, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool GEGLU = MODE == 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int lrow = lane >> 2, lchunk = lane & 3;
  const int nk = K >> 5;
  const int ntiles = tiles_m * tiles_n;
  const int nmy = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int ldd = GEGLU ? N / 2 : N;

  auto decode = [&](int vt, int &tm, int &tn) {
    const int t = xcd_remap(vt, ntiles);
    const int per_group = 4 * tiles_n, group = t / per_group, first_m = group * 4;
    const int gsz = min(tiles_m - first_m, 4), in_group = t - group * per_group;
    tn = in_group / gsz;
    tm = first_m + (in_group - tn * gsz);
  };

  // ---- producer: wave w issues pieces w, w+4, .. of the 16 A pieces and of the 16 B pieces of a K-step
  unsigned a_off[4], b_off[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int r = (wave + 4 * e) * 16 + lrow;
    a_off[e] = (unsigned)((r * K + (lchunk ^ swz4(r)) * 8) * 2);
    b_off[e] = a_off[e];
  }
  const char *a_base, *b_base;
  int s_kk = 0, s_tile = 0, p_slot = 0;           // stream position: K-step inside its tile, tile ordinal, ring slot
  auto set_stream_tile = [&](int ord) {
    int tm, tn;
    decode((int)blockIdx.x + ord * (int)gridDim.x, tm, tn);
    a_base = (const char *)A + (int64_t)tm * BM * K * 2;
    b_base = (const char *)W + (int64_t)tn * BN * K * 2;
  };
  set_stream_tile(0);
  auto dma_piece = [&](int e) {                   // e = 0..7: four A pieces, then four B pieces, of the stream's K-step
    if (BDIRECT && e >= 4) return;                // weights never touch LDS
    char *dst = smem + p_slot * SLOT + (e < 4 ? 0 : 16384) + (wave + 4 * (e & 3)) * 1024;
    glds16((e < 4 ? a_base + a_off[e & 3] : b_base + b_off[e & 3]), dst);   // (past the last tile: its first K-steps again, never read)
  };
  // BDIRECT: this wave's eight 16 x 32 weight fragments of a K-step come straight from L2 into registers
  // (global_load_dwordx4, saddr form: wave-uniform base per fragment + one per-lane offset), one K-step ahead
  const unsigned bd_voff = (unsigned)((fr * K + fq * 8) * 2);
  const char *bd_base = nullptr;                  // weights of the B-stream's tile, at its K-step
  int bd_kk = 0, bd_tile = 0;
  auto bd_set_tile = [&](int ord) {
    int tm, tn;
    decode((int)blockIdx.x + ord * (int)gridDim.x, tm, tn);
    bd_base = (const char *)W + ((int64_t)tn * BN + wn * 128) * K * 2;
  };
  auto bd_load = [&](f16x8 &dst, int i) {
    const char *sb = bd_base + (int64_t)i * 16 * K * 2;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(bd_voff), "s"(sb) : "memory");
  };
  auto bd_advance = [&]() {
    bd_base += 64;
    if (++bd_kk == nk) { bd_kk = 0; ++bd_tile; bd_set_tile(min(bd_tile, nmy - 1)); }
  };
  if constexpr (BDIRECT) bd_set_tile(0);
  auto dma_advance = [&]() {                      // after the 8 pieces of a K-step
    p_slot = (p_slot + 1) & (NSLOT - 1);
    a_base += 64; b_base += 64;
    if (++s_kk == nk) { s_kk = 0; ++s_tile; set_stream_tile(min(s_tile, nmy - 1)); }
  };
#pragma unroll
  for (int s = 0; s < DIST; ++s) {
#pragma unroll
    for (int e = 0; e < 8; ++e) dma_piece(e);
    dma_advance();
  }

  // ---- consumer
  const int rd_chunk = (fq ^ swz4(fr)) << 4;
  int offa[8], offb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    offa[j] = (wm * 128 + j * 16 + fr) * 64 + rd_chunk;
    offb[j] = 16384 + (wn * 128 + j * 16 + fr) * 64 + rd_chunk;
  }
  const int ocol = (fq & 1) * 16 + (fq >> 1) * 8;
  const __amdgpu_buffer_rsrc_t d_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void *)D, 0, MODE == 1 ? 0 : (int)min((int64_t)M * ldd * 2, (int64_t)0x7fffffff), 0x00020000);

  f32x4 acc[8][8];                                // [i: weight sub-tile][j: activation sub-tile]
  f16x8 fa[8], fb[2][8];
  constexpr int NOUT = GEGLU ? 16 : 32;           // 16-byte stores per wave and tile
  uint4 outq[NOUT];                               // MODE 2: the previous tile's packed outputs
  int out_row0 = 0, out_col0 = 0;
  bool have_out = false;

  // first K-step's fragments (exposed once per workgroup)
  constexpr int PPK = BDIRECT ? 4 : 8;            // LDS-DMA pieces per wave and K-step
  if constexpr (BDIRECT) {
#pragma unroll
    for (int i = 0; i < 8; ++i) bd_load(fb[0][i], i);
    bd_advance();
    wait_vm<0>();
  } else {
    wait_vm<2 * 8>();
  }
  __builtin_amdgcn_s_barrier();
  int read_slot = 0;
  {
    const char *s0 = smem;
    if constexpr (!BDIRECT) {
#pragma unroll
      for (int i = 0; i < 8; ++i) fb[0][i] = *(const f16x8 *)(s0 + offb[i]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[j] = *(const f16x8 *)(s0 + offa[j]);
  }
  int g = 0;                                      // global K-step ordinal of this workgroup

  // One K-step.  P: which fb buffer holds this K-step's weight fragments.  FIRST: accumulators start from zero.
  // young_extra: stores issued after the DMA of K-step g+2 that are still younger than the pieces of g+1 in the queue.
  auto kstep = [&](auto first_c, auto p_c, int extra) {
    constexpr bool FIRST = decltype(first_c)::value;
    constexpr int P = decltype(p_c)::value;
    const char *nxt = smem + ((read_slot + 1) & (NSLOT - 1)) * SLOT;     // slot of K-step g+1
    read_slot = (read_slot + 1) & (NSLOT - 1);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BDIRECT) {
      // this K-step's weight fragments were requested during the previous one, AHEAD of its four LDS-DMA pieces (and of
      // the stores that may have followed): everything older has landed once at most those remain
      if (extra == 0) wait_vm<4>(); else if (extra == 1) wait_vm<4 + NOUT>(); else wait_vm<0>();
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(fb[P][i]));
      SB();
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = j * 8 + i;
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[P][i], fa[j], FIRST ? zero : acc[i][j], 0, 0, 0);
        SB();
        if (q == 1) {
          // this wave's pieces of K-step g+1 have landed (g+2's may stay in flight, and whatever is younger: `extra`).
          // BDIRECT: the wait at the K-step's head (below) already covered them -- they are older than its weight loads.
          if constexpr (!BDIRECT) { if (extra == 0) wait_vm<8>(); else if (extra == 1) wait_vm<8 + NOUT>(); else wait_vm<0>(); }
          __builtin_amdgcn_s_barrier();
          SB();
        }
        // (after the workgroup's last K-step these read a slot nobody needs: no branch in the stream)
        if (q >= 2 && q <= 9) {
          if constexpr (BDIRECT) bd_load(fb[1 - P][q - 2], q - 2);
          else fb[1 - P][q - 2] = *(const f16x8 *)(nxt + offb[q - 2]);
          SB();
        }
        if (q >= 10 && ((q - 10) & 7) == 0) {
          const int jr = (q - 10) >> 3;              // rows 0..6 finished at q = 8*jr + 7 <= q - 3
          fa[jr] = *(const f16x8 *)(nxt + offa[jr]);
          SB();
        }
        if constexpr (BDIRECT) { if ((q & 15) == 14) { dma_piece(q >> 4); SB(); } }
        else { if ((q & 7) == 6) { dma_piece(q >> 3); SB(); } }
        if constexpr (MODE == 2) {
          // previous tile's outputs: one 16-byte store every 4th MFMA of the tile's first K-step
          if (FIRST && (q & 3) == 3 && have_out) {
            const int s = q >> 2, jj = s >> 1, oo = s & 1;      // NOUT = 16 = 8 rows x 2 pairs
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, outq[s]), d_rsrc,
                                                   (int)(((int64_t)(out_row0 + jj * 16) * ldd + out_col0 + oo * 32) * 2), 0, 2);
            SB();
          }
        }
      }
    }
    dma_advance();
    if constexpr (BDIRECT) bd_advance();
    fa[7] = *(const f16x8 *)(nxt + offa[7]);
    SB();
    ++g;
  };
  using T = std::true_type; using F = std::false_type;
  using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;

  for (int ti = 0; ti < nmy; ++ti) {
    int c_tm, c_tn;
    decode((int)blockIdx.x + ti * (int)gridDim.x, c_tm, c_tn);
    // stores of the previous tile: MODE 0 issues NOUT of them after the tile (younger than DMA g+2 for the next two K-steps);
    // MODE 2 issues them inside the first K-step (after its wait), so they are younger only in the second K-step
    const int x0 = (MODE == 0 && ti > 0) ? 1 : 0;
    const int x1 = BDIRECT ? 0 : (((MODE == 0 || MODE == 2) && ti > 0) ? 1 : 0);
    kstep(T{}, P0{}, x0);
    kstep(F{}, P1{}, x1);
    for (int kt = 2; kt < nk; kt += 2) {
      // (the K-step after the stores' two: the pieces of g+1 are OLDER than nothing but DMA g+2 again -- but the stores sit
      // between DMA g+1.. and must have been acknowledged: wait_vm<8> covers them, they are older than DMA g+2)
      kstep(F{}, P0{}, 0);
      kstep(F{}, P1{}, 0);
    }
    // ---- epilogue: pack (GEGLU: value * gelu(gate)), exchange with the neighbour 16 lanes away -> 8 channels per lane
    const int row0 = c_tm * BM + wm * 128 + fr;
    const int col0 = c_tn * (GEGLU ? BN / 2 : BN) + wn * (GEGLU ? 64 : 128) + ocol;
    if constexpr (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const f32x4 h0 = acc[4 * o][j], g0 = acc[4 * o + 1][j], h1 = acc[4 * o + 2][j], g1 = acc[4 * o + 3][j];
          const f32x2 ga = gelu2_f((f32x2){g0[0], g0[1]}), gb = gelu2_f((f32x2){g0[2], g0[3]});
          const f32x2 gc = gelu2_f((f32x2){g1[0], g1[1]}), gd = gelu2_f((f32x2){g1[2], g1[3]});
          uint4 out;
          out.x = pack_h2(h0[0] * ga[0], h0[1] * ga[1]); out.y = pack_h2(h0[2] * gb[0], h0[3] * gb[1]);
          out.z = pack_h2(h1[0] * gc[0], h1[1] * gc[1]); out.w = pack_h2(h1[2] * gd[0], h1[3] * gd[1]);
          swap16(out.x, out.z); swap16(out.y, out.w);
          outq[j * 2 + o] = out;
        }
      out_row0 = row0; out_col0 = col0; have_out = true;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const f32x4 va = acc[2 * o][j], vb = acc[2 * o + 1][j];
          uint4 out;
          out.x = pack_h2(va[0], va[1]); out.y = pack_h2(va[2], va[3]);
          out.z = pack_h2(vb[0], vb[1]); out.w = pack_h2(vb[2], vb[3]);
          swap16(out.x, out.z); swap16(out.y, out.w);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, out), d_rsrc,
                                                 (int)(((int64_t)(row0 + j * 16) * ldd + col0 + o * 32) * 2), 0, 2);
        }
    }
    SB();
  }
  if constexpr (MODE == 2) {
    if (have_out) {
#pragma unroll
      for (int s = 0; s < NOUT; ++s)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, outq[s]), d_rsrc,
                                               (int)(((int64_t)(out_row0 + (s >> 1) * 16) * ldd + out_col0 + (s & 1) * 32) * 2), 0, 2);
    }
  }
}

template <int MODE, bool BDIRECT = false>
float run(const f16 *A, const f16 *W, f16 *D, int M, int N, int K, int reps) {
  const int tiles_m = M / BM, tiles_n = N / BN, ntiles = tiles_m * tiles_n;
  const int grid = ntiles < 256 ? ntiles : 256;
  const size_t lds = NSLOT * SLOT;
  CHECK(hipFuncSetAttribute((const void *)w1_kernel<MODE, BDIRECT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((w1_kernel<MODE, BDIRECT>), dim3(grid), dim3(256), lds, 0, A, W, D, M, N, K, tiles_m, tiles_n);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((w1_kernel<MODE, BDIRECT>), dim3(grid), dim3(256), lds, 0, A, W, D, M, N, K, tiles_m, tiles_n);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char **argv) {
  int M = 258048, N = 2560, K = 320, reps = 20;
  if (argc >= 4) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
  if (argc >= 5) reps = atoi(argv[4]);
  if (M % BM || N % BN || K % 64 || (K / 32) % 2 || K / 32 < 4) { printf("need M %% 256 == 0, N %% 256 == 0, K %% 64 == 0, K >= 128\n"); return 1; }
  std::vector<f16> hA((size_t)M * K), hW((size_t)N * K);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f; };
  for (auto &v : hA) v = (f16)rnd();
  for (auto &v : hW) v = (f16)(rnd() * 0.1f);
  f16 *A, *W, *D;
  CHECK(hipMalloc(&A, hA.size() * 2)); CHECK(hipMalloc(&W, hW.size() * 2)); CHECK(hipMalloc(&D, (size_t)M * N * 2));
  CHECK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  const double flop = 2.0 * M * N * K;

  // ---- correctness of MODE 0 and MODE 2 on a sample
  for (int mode : {0, 2, 10}) {                    // 10: plain stores with the weights straight from L2
    CHECK(hipMemset(D, 0xff, (size_t)M * N * 2));
    if (mode == 0) run<0>(A, W, D, M, N, K, 1); else if (mode == 2) run<2>(A, W, D, M, N, K, 1); else run<0, true>(A, W, D, M, N, K, 1);
    const int ldd = mode == 2 ? N / 2 : N;
    std::vector<f16> hD((size_t)M * ldd);
    CHECK(hipMemcpy(hD.data(), D, hD.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0; int bad = 0;
    for (int t = 0; t < 4000; ++t) {
      s = s * 1664525u + 1013904223u; const int m = (s >> 4) % M;
      s = s * 1664525u + 1013904223u; const int c = (s >> 4) % ldd;
      double ref;
      auto dot = [&](int n) { double a = 0; for (int k = 0; k < K; ++k) a += (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k]; return a; };
      if (mode == 2) {   // value / gate interleaved in blocks of 16 columns: out c <- (32*(c/16) + c%16, +16)
        const int nv = 32 * (c / 16) + c % 16;
        const double v = dot(nv), gt = dot(nv + 16);
        ref = v * 0.5 * gt * (1.0 + erf(gt / sqrt(2.0)));
      } else ref = dot(c);
      const double got = (double)hD[(size_t)m * ldd + c];
      const double err = fabs(got - ref) / (fabs(ref) + 0.05);
      if (err > worst) worst = err;
      if (err > 2e-2) ++bad;
    }
    printf("mode %d check: worst rel err %.3e, %d / 4000 beyond 2e-2 %s\n", mode, worst, bad, bad ? "FAILED" : "ok");
    if (bad) return 2;
  }
  for (int round = 0; round < 3; ++round) {
    const float t0 = run<0>(A, W, D, M, N, K, reps), t1 = run<1>(A, W, D, M, N, K, reps), t2 = run<2>(A, W, D, M, N, K, reps);
    const float t3 = run<0, true>(A, W, D, M, N, K, reps), t4 = run<1, true>(A, W, D, M, N, K, reps);
    printf("M %d N %d K %d: plain+stores %.1f us (%.0f TFLOP/s) | K loop only %.1f us (%.0f) | GEGLU, stores under next tile %.1f us (%.0f)"
           " || weights from L2 to registers: plain+stores %.1f us (%.0f) | K loop only %.1f us (%.0f)\n",
           M, N, K, 1e3 * t0, flop / t0 * 1e-9, 1e3 * t1, flop / t1 * 1e-9, 1e3 * t2, flop / t2 * 1e-9,
           1e3 * t3, flop / t3 * 1e-9, 1e3 * t4, flop / t4 * 1e-9);
  }
  return 0;
}

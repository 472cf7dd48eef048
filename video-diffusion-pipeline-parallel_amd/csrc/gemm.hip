// Implicit-GEMM core for gfx950: D[M][N] = epilogue(sum_tap A_tap[M][Cin] * W[N][tap*Cin + :]^T).
//
// One kernel serves every dense contraction of the SVD UNet step:
//   * SP_A_LINEAR    – nn.Linear / 1x1 convolution (row m of A is row m of the token matrix)
//   * SP_A_CONV3X3   – 3x3 convolution, pad 1, stride 1|2, optional fused nearest x2 upsample;
//                      the 9 taps are gathered straight from the NHWC tensor (no im2col buffer)
//   * SP_A_TEMPORAL3 – (3,1,1) convolution over frames; taps are rows m-hw, m, m+hw
//
// Structure: tile BM x BN x 64 per workgroup of WM x WN waves, three shapes chosen on the host:
//     256 x 128, 8 waves (4x2), 3-deep LDS ring  – large-M contractions (the bulk of the FLOPs); the
//                 LDS-DMA prefetch runs TWO K-steps ahead and stays in flight across the barrier behind a
//                 counted s_waitcnt vmcnt(L) (raw s_barrier, never __syncthreads in the loop)
//     128 x {128,160,64}, 4 waves (2x2), 2-deep ring, 2 workgroups per CU – N = 320/960/1920 and mid sizes
//      64 x 64, 4 waves, 3-deep ring – the 2016-row level, so that the grid still fills 256 CUs
//   A and W tiles are brought in by global_load_lds_dwordx4
//   (LDS-DMA, 1 KiB per wave instruction = 8 rows x 128 B); the 16-byte
//   chunk index is XOR-swizzled on the *source* address ((row>>1)&7) and un-swizzled on the
//   ds_read_b128, which makes the 16x16x32 operand reads bank-conflict free; MFMA
//   v_mfma_f32_16x16x32_f16 with the weight tile as the row operand so that each lane ends up with 4
//   consecutive output channels (8-byte LDS staging writes, 16-byte coalesced global stores).
//   Out-of-image taps / rows past M read a zero page, so no branch sits in the main loop.
//   The workgroup -> tile map is XCD-aware: the 8 XCDs each walk a contiguous run of tiles with N
//   fastest, so an A tile is fetched from HBM once per XCD L2 and re-used by all N tiles.
#include <cstring>
#include "gemm_args.h"
#include <stdlib.h>

using namespace spgemm;

namespace {


template <int BM, int BN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(WM *WN * 64, 2) void gemm_f16_kernel(const GemmArgs p) {
  constexpr int NW = WM * WN;
  constexpr int THREADS = NW * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;   // wave tile
  constexpr int TN = WTN / 16;         // 16-row weight sub-tiles per wave
  constexpr int TM = WTM / 16;         // 16-col activation sub-tiles per wave
  constexpr int A_BYTES = BM * BK * 2;
  constexpr int B_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int A_LOADS = BM / 8 / NW; // glds (8-row pieces) per thread for the A tile
  constexpr int B_LOADS = BN / 8 / NW; // ... and for the W tile
  constexpr int L = A_LOADS + B_LOADS; // LDS-DMA instructions per wave per K-step
  static_assert(A_LOADS * 8 * NW == BM && B_LOADS * 8 * NW == BN, "tile rows must split evenly over waves");
  static_assert(STAGES == 2 || STAGES == 3, "2- or 3-deep ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware, L2-friendly rasterisation: each XCD walks a contiguous run of tile ids, and ids are grouped
  // GM M-tiles x all N-tiles (column by column), so the ~32-64 workgroups an XCD runs at once form a
  // GM x (conc/GM) patch: per K-step the patch shares GM A-tiles and conc/GM W-tiles through that XCD's L2.
  constexpr int GM = BM >= 256 ? 4 : 8;
  const int nwg = p.tiles_m * p.tiles_n;
  const int t = xcd_remap(blockIdx.x, nwg);
  const int per_group = GM * p.tiles_n;
  const int group = t / per_group;
  const int first_m = group * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int in_group = t - group * per_group;
  const int tile_n = in_group / gsz;
  const int tile_m = first_m + (in_group - tile_n * gsz);

  // ---------------------------------------------------------------- per-lane gather state
  const int lrow = lane >> 3;                 // row within the 8-row glds piece
  const int lchunk = lane & 7;                // 16-byte chunk this lane lands in (LDS side)
  const int cpt = p.cin >> 6;                 // K-steps per tap

  int a_i0[A_LOADS], a_i1[A_LOADS], a_i2[A_LOADS];   // conv: img, iy0, ix0 | temporal: frame, -, m | -, -, m
  bool a_in[A_LOADS];
  int schunk_a[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int r = (wave * A_LOADS + i) * 8 + lrow;
    const int m = tile_m * BM + r;
    a_in[i] = m < p.m;
    schunk_a[i] = (lchunk ^ ((r >> 1) & 7)) * 8;  // source chunk (halves) for this LDS slot
    if (p.mode == SP_A_CONV3X3) {
      const int per_img = p.hout * p.wout;
      const int img = m / per_img;
      const int rem = m - img * per_img;
      const int oy = rem / p.wout;
      a_i0[i] = img;
      a_i1[i] = oy * p.stride - 1;
      a_i2[i] = (rem - oy * p.wout) * p.stride - 1;
    } else if (p.mode == SP_A_TEMPORAL3) {
      a_i0[i] = (int)((m / p.hw) % p.frames);
      a_i1[i] = 0;
      a_i2[i] = m;
    } else {
      a_i0[i] = a_i1[i] = 0;
      a_i2[i] = m;
    }
  }
  const f16 *aptr[A_LOADS];
  int astep[A_LOADS];
  auto set_tap = [&](int tap) {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      int64_t row = -1;
      if (a_in[i]) {
        if (p.mode == SP_A_CONV3X3) {
          const int ky = tap / 3, kx = tap - ky * 3;
          const int iy = a_i1[i] + ky, ix = a_i2[i] + kx;
          const int hv = p.hin << p.ups, wv = p.win << p.ups;
          if (iy >= 0 && iy < hv && ix >= 0 && ix < wv)
            row = ((int64_t)a_i0[i] * p.hin + (iy >> p.ups)) * p.win + (ix >> p.ups);
        } else if (p.mode == SP_A_TEMPORAL3) {
          const int f = a_i0[i] + tap - 1;
          if (f >= 0 && f < p.frames) row = (int64_t)a_i2[i] + (int64_t)(tap - 1) * p.hw;
        } else {
          row = a_i2[i];
        }
      }
      if (row >= 0) {
        aptr[i] = p.a + row * p.lda + schunk_a[i];
        astep[i] = BK;
      } else {
        aptr[i] = (const f16 *)(p.zero + lchunk * 16);
        astep[i] = 0;
      }
    }
  };

  const f16 *bptr[B_LOADS];
#pragma unroll
  for (int j = 0; j < B_LOADS; ++j) {
    const int r = (wave * B_LOADS + j) * 8 + lrow;
    const int n = tile_n * BN + r;
    bptr[j] = p.w + (int64_t)n * p.k + (lchunk ^ ((r >> 1) & 7)) * 8;
  }

  auto stage = [&](int buf) {
    char *sa = smem + buf * STAGE;
    char *sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      glds16(aptr[i], sa + (wave * A_LOADS + i) * 1024);
      aptr[i] += astep[i];
    }
#pragma unroll
    for (int j = 0; j < B_LOADS; ++j) {
      glds16(bptr[j], sb + (wave * B_LOADS + j) * 1024);
      bptr[j] += BK;
    }
  };

  // ---------------------------------------------------------------- main loop
  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.k >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  // fragment byte offsets inside a stage (row*128 + swizzled chunk), chunk = ks*4 + fq
  int offw[TN], offa[TM];
#pragma unroll
  for (int i = 0; i < TN; ++i) offw[i] = (wn * WTN + i * 16 + fr) * 128;
#pragma unroll
  for (int j = 0; j < TM; ++j) offa[j] = (wm * WTM + j * 16 + fr) * 128;
  // (row>>1)&7 of the fragment rows: rows are base16 + fr with base16 % 16 == 0
  const int swz = (fr >> 1) & 7;

  // K-steps are staged through a STAGES-deep ring; `staged` counts K-steps whose LDS-DMA has been issued.
  int staged = 0, in_tap = 0, tap = 0;
  auto stage_next = [&]() {
    if (in_tap == cpt) { ++tap; in_tap = 0; set_tap(tap); }
    stage(staged % STAGES);
    ++staged; ++in_tap;
  };
  set_tap(0);
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) stage_next();
  // K-step 0 must have landed; with a 3-deep ring K-step 1 may stay in flight
  if (STAGES == 3 && nk > 1) wait_vm_lgkm<L>(); else wait_vm_lgkm<0>();
  __builtin_amdgcn_s_barrier();

  if constexpr (NW == 8 && STAGES == 3) {
    // ---- ping-pong schedule (8 waves = two waves per SIMD; waves w and w+4 share a SIMD) ------------------
    // Every K-step is split into a READ phase (issue the LDS-DMA prefetch for K-step kt+2, pull all 16 operand
    // fragments of K-step kt into registers) and a COMPUTE phase (32 MFMAs from registers), separated by
    // workgroup barriers.  Waves 4-7 run half a K-step behind waves 0-3 (one extra barrier up front), so on
    // every SIMD one wave is always in its MFMA phase while its partner reads LDS: the matrix pipe does not idle
    // during operand fetch.  Both halves retire K-step kt+1's DMA (counted vmcnt, K-step kt+2 stays in flight)
    // before the barrier that precedes the first read of it.
    const bool late = wave >= 4;
    if (late) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 2 < nk;
      if (more) stage_next();
      const char *sa = smem + (kt % 3) * STAGE;
      const char *sb = sa + A_BYTES;
      f16x8 fw[2][TN], fa[2][TM];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int coff = ((ks * 4 + fq) ^ swz) << 4;
#pragma unroll
        for (int i = 0; i < TN; ++i) fw[ks][i] = *(const f16x8 *)(sb + offw[i] + coff);
#pragma unroll
        for (int j = 0; j < TM; ++j) fa[ks][j] = *(const f16x8 *)(sa + offa[j] + coff);
      }
      if (late) { if (more) wait_vm_lgkm<L>(); else wait_vm_lgkm<0>(); }
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[ks][i], fa[ks][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      if (!late) { if (more) wait_vm_lgkm<L>(); else wait_vm_lgkm<0>(); }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!late) __builtin_amdgcn_s_barrier();
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + STAGES - 1 < nk;
      if (more) stage_next();
      const char *sa = smem + (kt % STAGES) * STAGE;
      const char *sb = sa + A_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int coff = ((ks * 4 + fq) ^ swz) << 4;
        f16x8 fw[TN], fa[TM];
#pragma unroll
        for (int i = 0; i < TN; ++i) fw[i] = *(const f16x8 *)(sb + offw[i] + coff);
#pragma unroll
        for (int j = 0; j < TM; ++j) fa[j] = *(const f16x8 *)(sa + offa[j] + coff);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
      }
      // K-step kt+1 must be in LDS before the next iteration reads it; the K-step issued this iteration
      // (3-deep ring) stays in flight across the barrier.  lgkmcnt(0): this wave's reads of slot kt are done
      // before any wave may overwrite it.
      if (STAGES == 3 && more) wait_vm_lgkm<L>(); else wait_vm_lgkm<0>();
      __builtin_amdgcn_s_barrier();
    }
  }

  // ---------------------------------------------------------------- epilogue
  // acc[i][j][r]: n = tile_n*BN + wn*WTN + i*16 + 4*fq + r ; m = tile_m*BM + wm*WTM + j*16 + fr
  constexpr int BNO_FULL = BN;
  const int bno = p.geglu ? BN / 2 : BNO_FULL;   // output columns of this tile
  const int ldc = bno + 8;                        // halves, padded
  f16 *sc = (f16 *)smem;

  if (!p.geglu) {
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int nl = wn * WTN + i * 16 + 4 * fq;
      const int n = tile_n * BN + nl;
      f32x4 b = {0.f, 0.f, 0.f, 0.f}, cs = b;
      if (p.bias) b = *(const f32x4 *)(p.bias + n);
      if (p.ln_stats) cs = *(const f32x4 *)(p.ln_colsum + n);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int ml = wm * WTM + j * 16 + fr;
        f32x4 v = acc[i][j] + b;
        if (p.ln_stats) {       // folded LayerNorm: rstd*(acc - mean*colsum) + bias
          const int64_t mm = min((int64_t)tile_m * BM + ml, (int64_t)p.m - 1);
          const float2 st = *(const float2 *)(p.ln_stats + mm * 2);
          v = (acc[i][j] - st.x * cs) * st.y + b;
        }
        if (p.bias2) {
          const int64_t m = (int64_t)tile_m * BM + ml;
          const int64_t brow = m < p.m ? m / p.bias2_rows : 0;
          v += *(const f32x4 *)(p.bias2 + brow * p.ldb2 + n);
        }
        v *= p.oscale;
        f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
        *(f16x4 *)(sc + ml * ldc + nl) = h;
      }
    }
  } else {
    if constexpr (TN % 2 == 0 && WTN % 32 == 0) {
#pragma unroll
      for (int i = 0; i < TN; i += 2) {
        const int nl = wn * WTN + i * 16 + 4 * fq;   // h rows; gate rows are nl + 16
        const int n = tile_n * BN + nl;
        f32x4 bh = {0.f, 0.f, 0.f, 0.f}, bg = bh, ch = bh, cg = bh;
        if (p.bias) {
          bh = *(const f32x4 *)(p.bias + n);
          bg = *(const f32x4 *)(p.bias + n + 16);
        }
        if (p.ln_stats) {
          ch = *(const f32x4 *)(p.ln_colsum + n);
          cg = *(const f32x4 *)(p.ln_colsum + n + 16);
        }
        const int ol = (wn * WTN + i * 16) / 2 + 4 * fq;  // output column within tile
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int ml = wm * WTM + j * 16 + fr;
          f32x4 hv = acc[i][j] + bh, gv = acc[i + 1][j] + bg;
          if (p.ln_stats) {
            const int64_t mm = min((int64_t)tile_m * BM + ml, (int64_t)p.m - 1);
            const float2 st = *(const float2 *)(p.ln_stats + mm * 2);
            hv = (acc[i][j] - st.x * ch) * st.y + bh;
            gv = (acc[i + 1][j] - st.x * cg) * st.y + bg;
          }
          f16x4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = (f16)(p.oscale * hv[r] * gelu_f(gv[r]));
          *(f16x4 *)(sc + ml * ldc + ol) = h;
        }
      }
    }
  }
  __syncthreads();

  const int cpr = bno >> 3;                        // 16-byte chunks per output row
  const int ncols_total = p.geglu ? p.n / 2 : p.n;
  const int nstore = p.n_store > 0 ? p.n_store : ncols_total;
  for (int idx = tid; idx < BM * cpr; idx += THREADS) {
    const int r = idx / cpr, c = idx - r * cpr;
    const int64_t m = (int64_t)tile_m * BM + r;
    if (m >= p.m) continue;
    const int col = tile_n * bno + c * 8;
    if (col >= nstore) continue;
    f16x8 v = *(const f16x8 *)(sc + r * ldc + c * 8);
    if (p.eul_out) {
      // guidance mix + Euler update of the four latent channels of row m = (b, f, pixel) (same arithmetic, same
      // roundings as sp_euler_step_f16 on the eps rows this epilogue would otherwise have written)
      if (col != 0) continue;
      const int64_t pix = m % p.eul_hw, bf = m / p.eul_hw;
      const int f = (int)(bf % p.eul_frames);
      const int64_t b = bf / p.eul_frames;
      f16x4 u4 = {v[0], v[1], v[2], v[3]};
      float g = 1.f;
      if (p.eul_u) { u4 = *(const f16x4 *)(p.eul_u + m * p.eul_ldu); g = p.eul_gs[f]; }
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        const int64_t a = ((b * 4 + ch) * p.eul_frames + f) * p.eul_hw + pix;
        const float x = (float)p.eul_lat[a];
        float e = (float)v[ch];
        if (p.eul_u) {
          const f16 gh = (f16)g;
          const f16 diff = (f16)((float)v[ch] - (float)u4[ch]);
          const f16 prod = (f16)((float)gh * (float)diff);
          e = (float)(f16)((float)u4[ch] + (float)prod);
        }
        const float x0 = e * p.eul_c_out + x * p.eul_c_skip;
        const float d = (x - x0) * p.eul_inv_sigma;
        p.eul_out[a] = (f16)(x + d * p.eul_dt);
      }
      continue;
    }
    if (p.res1 || p.res2) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
      if (p.res1) {
        const f16x8 q = *(const f16x8 *)(p.res1 + m * p.ldr1 + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += p.r1scale * (float)q[e];
      }
      if (p.res2) {
        const f16x8 q = *(const f16x8 *)(p.res2 + m * p.ldr2 + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += p.r2scale * (float)q[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (f16)f[e];
    }
    if (col + 8 <= nstore) {
      *(f16x8 *)(p.d + m * p.ldd + col) = v;
    } else {
      for (int e = 0; e < nstore - col; ++e) p.d[m * p.ldd + col + e] = v[e];
    }
  }
}

template <int BM, int BN, int WM, int WN, int STAGES>
int launch(GemmArgs &a, hipStream_t s) {
  spgemm::note_kernel("gemm_f16_kernel<%d, %d, %d, %d, %d>", BM, BN, WM, WN, STAGES);
  constexpr size_t lds = (size_t)STAGES * (BM + BN) * BK * 2;
  static_assert((size_t)BM * (BN + 8) * 2 <= lds, "epilogue staging tile must fit in the ring");
  static bool attr_set[SP_MAX_DEVICES] = {};
  if (int rc = sp_ensure_dyn_lds((const void *)gemm_f16_kernel<BM, BN, WM, WN, STAGES>, (int)lds, attr_set, "sp_gemm_f16"))
    return rc;
  a.tiles_m = (a.m + BM - 1) / BM;
  a.tiles_n = a.n / BN;
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, WM, WN, STAGES>), dim3(a.tiles_m * a.tiles_n), dim3(WM * WN * 64),
                     lds, s, a);
  SP_CHECK_LAUNCH("sp_gemm_f16");
  return SP_OK;
}

}  // namespace

namespace {

// Second half of a split-K contraction: sums the fp32 slabs of the K slices in slice order (deterministic) and applies
// the whole sp_gemm_f16 epilogue (bias, bias2 row, folded LayerNorm, output scale, GEGLU, residuals, n_store).  One
// thread per 8 output columns of one row.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p, int nout, int nstore) {
  const int cpr = (nstore + 7) >> 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)p.m * cpr) return;
  const int64_t m = idx / cpr;
  const int col = (int)(idx - m * cpr) * 8;
  const int64_t slab = (int64_t)p.m * p.n;
  // accumulator columns behind output columns col..col+7: the same columns, or for GEGLU the value / gate rows of the
  // interleaved weight (blocks of 16: value 32q + r, gate 32q + 16 + r for output column 16q + r)
  const int nv = p.geglu ? 32 * (col >> 4) + (col & 15) : col;
  float v[8], g[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = g[e] = 0.f;
  for (int sl = 0; sl < p.ksplit; ++sl) {
    const float *src = p.partial + sl * slab + m * p.n + nv;
    const f32x4 a0 = *(const f32x4 *)src, a1 = *(const f32x4 *)(src + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] += a0[e]; v[e + 4] += a1[e]; }
    if (p.geglu) {
      const f32x4 b0 = *(const f32x4 *)(src + 16), b1 = *(const f32x4 *)(src + 20);
#pragma unroll
      for (int e = 0; e < 4; ++e) { g[e] += b0[e]; g[e + 4] += b1[e]; }
    }
  }
  float mean = 0.f, rstd = 1.f;
  if (p.ln_stats) { mean = p.ln_stats[m * 2]; rstd = p.ln_stats[m * 2 + 1]; }
  const int64_t brow = p.bias2 ? m / p.bias2_rows : 0;
  auto finish = [&](float acc, int n) {
    float b = p.bias ? p.bias[n] : 0.f;
    if (p.bias2) b += p.bias2[brow * p.ldb2 + n];
    return p.ln_stats ? (acc - mean * p.ln_colsum[n]) * rstd + b : acc + b;
  };
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float f;
    if (p.geglu) f = p.oscale * finish(v[e], nv + e) * gelu_f(finish(g[e], nv + 16 + e));
    else f = p.oscale * finish(v[e], nv + e);
    if (col + e < nstore) {
      if (p.res1) f += p.r1scale * (float)p.res1[m * p.ldr1 + col + e];
      if (p.res2) f += p.r2scale * (float)p.res2[m * p.ldr2 + col + e];
    }
    o[e] = (f16)f;
  }
  if (col + 8 <= nstore) {
    *(f16x8 *)(p.d + m * p.ldd + col) = o;
  } else {
    for (int e = 0; e < nstore - col; ++e) p.d[m * p.ldd + col + e] = o[e];
  }
}

// split-K plan for few-row contractions: number of K slices (0 = do not split) for 256 x 256 tiles.  Up to 6,144 rows
// (24 x 5 tiles at 1,280 columns still leave half the chip idle): the 2,016-row level of one video and the 4,032-row
// level of a micro-batch of two.
constexpr int SPLITK_MAX_ROWS = 6144;
int splitk_slices(const sp_gemm_desc *d, int taps, bool forced) {
  if (d->n % 256 || d->m > SPLITK_MAX_ROWS || d->euler_out) return 0;
  const int nk = taps * d->cin / 32;
  // K >= 8192 (the 3x3 convolutions over 1280 / 2560 channels: 100 -> 74 us, 203 -> 120 us).  Slab writes + the reduce
  // kernel cost ~20 us per call, which at K = 3840 / 5120 / 5760 (temporal convolution, FF2, 640-channel 3x3) is what
  // the larger tiles save: measured 45 vs 38, 54 vs 50, 52 vs 51 us against 64 x 64 tiles.
  if (!forced && nk < 256) return 0;
  const int tiles = ((d->m + 255) / 256) * (d->n / 256);
  int s = 256 / tiles;                                      // fill the 256 CUs once
  const int min_steps = forced ? 4 : 16;
  if (s > nk / min_steps) s = nk / min_steps;
  if (s > 8) s = 8;
  return s >= 2 ? s : 0;
}

// Kernel-family override for tests and micro-benchmarks (sp_gemm_set_route); 0 = automatic everywhere.
int g_route = 0, g_route_bm = 0, g_route_bn = 0;

// ceil(tiles / 256 CUs) rounds of bm x bn tiles: the busiest CU's share of the output
double makespan(int m, int n, int bm, int bn) {
  const int tiles = ((m + bm - 1) / bm) * (n / bn);
  return (double)((tiles + 255) / 256) * bm * bn;
}

// (mean, rstd) of rows whose per-tile (sum, sum of squares) a two-tile contraction left in `part` ([m][tiles][2])
__global__ __launch_bounds__(256) void ln_part_finalize_kernel(const float *__restrict__ part, float *__restrict__ out, int m,
                                                               int tiles, int n, float eps) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= m) return;
  float sm = 0.f, sq = 0.f;
  for (int t = 0; t < tiles; ++t) {
    const float2 v = *(const float2 *)(part + ((int64_t)row * tiles + t) * 2);
    sm += v.x; sq += v.y;
  }
  const float inv = 1.0f / (float)n, mean = sm * inv;
  float var = sq * inv - mean * mean;
  if (var < 0.f) var = 0.f;
  *(float2 *)(out + (int64_t)row * 2) = make_float2(mean, rsqrtf(var + eps));
}

int dispatch(GemmArgs &a, const sp_gemm_desc *d, hipStream_t s) {
  const bool n128 = d->n % 128 == 0;
  const bool ok256 = d->n % 256 == 0, ok320 = d->n % 320 == 0 && !d->geglu;
  const int route = g_route;

  // ---- output-row LayerNorm statistics live in the ping-pong epilogue (one tile = whole rows)
  // per-row-group weights: a tile must lie inside one group (192-row tiles need groups of a multiple of 192 rows, ...)
  auto group_ok = [&](int bm) { return a.w_group_rows == 0 || a.w_group_rows % bm == 0; };
  if (a.gn_part || a.a2)                       // per-tile column sums / the extra linear tap live in the 256-row ping-pong tiles
    return launch_pp(a, 256, ok320 ? 320 : 256, s);
  if (a.ln_out) {
    const int bn = d->n % 320 == 0 ? 320 : 256;
    int bm = makespan(d->m, d->n, 192, bn) * 1.06 < makespan(d->m, d->n, 256, bn) ? 192 : 256;
    if (!group_ok(bm)) bm = group_ok(256) ? 256 : 192;
    SP_REQUIRE(group_ok(bm), "sp_gemm_f16: w_group_rows=%lld fits neither 256- nor 192-row tiles", (long long)a.w_group_rows);
    if (int rc = launch_pp(a, bm, bn, s)) return rc;
    if (a.ln_part) {
      SP_CLEAR_STALE_ERROR();
      hipLaunchKernelGGL(ln_part_finalize_kernel, dim3((d->m + 255) / 256), dim3(256), 0, s, (const float *)a.ln_part, a.ln_out,
                         d->m, d->n / bn, d->n, a.ln_out_eps);
      SP_CHECK_LAUNCH("sp_gemm_f16(ln_out finalize)");
      spgemm::note_kernel_suffix(" + ln_part_finalize_kernel");
    }
    return SP_OK;
  }

  if (a.w_group_rows) {                       // ping-pong kernels only (the others walk one weight matrix)
    const int bn = ok320 ? 320 : 256;
    int bm = 0;
    const int bms[3] = {256, 192, 128};
    for (int i = 0; i < 3 && !bm; ++i)
      if (group_ok(bms[i]) && !(bms[i] == 128 && bn != 256)) bm = bms[i];
    SP_REQUIRE(bm, "sp_gemm_f16: w_group_rows=%lld fits no tile height for n=%d", (long long)a.w_group_rows, d->n);
    return launch_pp(a, bm, bn, s);
  }

  // ---- persistent-stream tiles (gemm_ps.hip): linear contractions with several tiles per CU, where launch gap,
  // index setup, first-operand latency and the LDS-staged epilogue of a one-tile workgroup are a large share
  if ((route == 0 || route == 3) && (ok256 || ok320 || d->n % 192 == 0)) {
    int bm = 0, bn = 0;
    double best = 0.0;
    const int cand[4][2] = {{256, 256}, {192, 256}, {128, 320}, {256, 192}};
    for (int c = 0; c < 4; ++c) {
      const int cbm = cand[c][0], cbn = cand[c][1];
      if ((cbn == 256 && !ok256) || (cbn == 320 && !ok320) || (cbn == 192 && (d->n % 192 || d->geglu))) continue;
      // 256 x 192 tiles (round 4; widths that are multiples of 192 but not of 256: the fused Q/K/V projections of the two
      // outer levels): built and verified, never chosen automatically.  With the SAME operand buffers launch after launch
      // (Infinity-Cache-resident) they beat the ping-pong kernel by 25 % at 258,048 x 960 x 320 (266.6 -> 212.6 us); with
      // operands that a launch does not find in the cache (tools/bench_routes.py COLD=6), as inside a forward, they lose
      // 6 % there and 14 % at 64,512 x 1,920 x 640, and the in-situ A/B reads -0.2 % / 0 (profiles/r04_ps256x192.txt)
      if (cbn == 192 && !(route == 3 && g_route_bn == 192)) continue;
      if (route == 3 && g_route_bn == 192 && cbn != 192) continue;
      // 128 x 320 tiles: built and verified (round 3), never chosen automatically -- 1.4x the LDS-DMA pieces per FLOP of
      // a 256-row tile; measured 0-25 % slower than the ping-pong kernel's 256 x 320 tiles on every N = 320 k shape of
      // the two outer levels
      if (cbm == 128 && !(route == 3 && g_route_bm == 128)) continue;
      a.tiles_m = (d->m + cbm - 1) / cbm;
      a.tiles_n = d->n / cbn;
      if (!spgemm::ps_supported(a, cbm, cbn)) continue;
      double t = makespan(d->m, d->n, cbm, cbn);
      if (cbm == 192) t *= 1.10;            // smaller tiles move more operand bytes per FLOP (measured 8-10 %)
      if (!bm || t < best) { best = t; bm = cbm; bn = cbn; }
    }
    if (bm) {
      const int tiles = ((d->m + bm - 1) / bm) * (d->n / bn);
      if (route == 3 || tiles >= 448) return launch_ps(a, bm, bn, s);
    }
  }

  // ---- few rows, long K (the 2016-row level's 3x3 convolutions): split-K on 256 x 256 tiles when
  // the caller provides the fp32 workspace; 64 x 64 tiles move four times the operand bytes per FLOP
  if ((route == 0 || route == 4) && d->workspace) {
    const int sk = splitk_slices(d, a.taps, route == 4);
    if (sk && d->workspace_bytes >= (size_t)sk * d->m * d->n * sizeof(float)) {
      a.ksplit = sk;
      a.partial = (float *)d->workspace;
      if (int rc = launch_pp(a, 256, 256, s)) return rc;
      const int nout = d->geglu ? d->n / 2 : d->n, nstore = d->n_store > 0 ? d->n_store : nout;
      const int64_t work = (int64_t)d->m * ((nstore + 7) / 8);
      SP_CLEAR_STALE_ERROR();
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, a, nout, nstore);
      SP_CHECK_LAUNCH("sp_gemm_f16(split-K reduce)");
      return SP_OK;
    }
  }
  if (d->m <= 2560 && route != 2) {   // few rows (the 2016-row level): small tiles so the grid still covers 256 CUs
    if (d->n >= 3840 && n128) return launch<128, 128, 2, 2, 2>(a, s);
    if (route == 1 && g_route_bm == 128) return launch<128, 64, 2, 2, 2>(a, s);     // (micro-benchmarks)
    return launch<64, 64, 2, 2, 3>(a, s);
  }
  // ---- a few thousand rows (the 4,032-row level of a micro-batch of two): where the large tiles would leave more than
  // a third of the CUs without a workgroup, 128 x 128 tiles (two workgroups per CU) win: measured at 4,032 x 1,280:
  // (3,1,1) convolution 54.5 vs 67.2 us, linear + residual 26.8 vs 31.1, FF2 (K = 5,120) 72.6 vs 91.3
  if (route == 0 && d->m <= SPLITK_MAX_ROWS && n128 && ok256 &&
      ((d->m + 191) / 192) * (d->n / 256) < 160)
    return launch<128, 128, 2, 2, 2>(a, s);
  // ---- large ping-pong tiles (gemm_pp.hip) for every N that is a multiple of 256 or 320
  if (route != 1 && (ok256 || ok320)) {
    // Pick the (BM, BN) whose last round of workgroups wastes the fewest of the 256 CUs (one workgroup per CU);
    // larger tiles win ties because they move fewer operand bytes per FLOP.
    int bm = 256, bn = ok256 ? 256 : 320;
    double best = -1.0;
    const int bms[2] = {256, 192}, bns[2] = {256, 320};
    for (int bi = 0; bi < 2; ++bi)
      for (int ni = 0; ni < 2; ++ni) {
        if ((bns[ni] == 256 && !ok256) || (bns[ni] == 320 && !ok320)) continue;
        const int blocks = ((d->m + bms[bi] - 1) / bms[bi]) * (d->n / bns[ni]);
        double score = (double)blocks / (((blocks + 255) / 256) * 256.0);
        if (bms[bi] == 192) score -= 0.06;
        if (bns[ni] == 320) score -= 0.01;
        if (score > best) { best = score; bm = bms[bi]; bn = bns[ni]; }
      }
    if (route == 2) {
      if ((g_route_bn == 256 && ok256) || (g_route_bn == 320 && ok320)) bn = g_route_bn;
      if (g_route_bm == 192 || g_route_bm == 256 || (g_route_bm == 128 && bn == 256)) bm = g_route_bm;
    }
    return launch_pp(a, bm, bn, s);
  }
  if (n128 && d->m >= 4096) return launch<256, 128, 4, 2, 3>(a, s);
  if (n128) return launch<128, 128, 2, 2, 2>(a, s);
  if (d->n % 160 == 0) return launch<128, 160, 2, 2, 2>(a, s);
  return launch<128, 64, 2, 2, 2>(a, s);
}

}  // namespace

namespace spgemm {
static thread_local char g_last_kernel[96] = "";
void note_kernel(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_kernel, sizeof(g_last_kernel), fmt, ap);
  va_end(ap);
}
void note_kernel_suffix(const char *suffix) {
  const size_t have = strlen(g_last_kernel);
  snprintf(g_last_kernel + have, sizeof(g_last_kernel) - have, "%s", suffix);
}
}  // namespace spgemm

extern "C" const char *sp_gemm_last_kernel(void) { return spgemm::g_last_kernel; }

extern "C" size_t sp_gemm_workspace_bytes(const sp_gemm_desc *d) {
  if (!d || d->m <= 0 || d->n <= 0 || d->cin <= 0) return 0;
  const int taps = d->mode == SP_A_CONV3X3 ? 9 : d->mode == SP_A_TEMPORAL3 ? 3 : 1;
  int sk = splitk_slices(d, taps, false);
  const int forced = splitk_slices(d, taps, true);
  if (forced > sk) sk = forced;                             // (enough for a forced split in tests as well)
  size_t need = (size_t)sk * d->m * d->n * sizeof(float);
  // ln_out over rows of several tiles (n = 512 ... 1280): the per-tile (sum, sum of squares) pairs meet here
  const int ltiles = d->n % 320 == 0 ? d->n / 320 : d->n / 256;
  const size_t ln_part = (d->ln_out && ltiles > 1) ? (size_t)d->m * ltiles * 2 * sizeof(float) : 0;
  return need > ln_part ? need : ln_part;
}

extern "C" int sp_gemm_set_route(int route, int bm, int bn) {
  SP_REQUIRE(route >= 0 && route <= 4,
             "sp_gemm_set_route: route %d (0 auto, 1 small tiles, 2 ping-pong, 3 persistent-stream, 4 split-K)", route);
  SP_REQUIRE(bm == 0 || bm == 128 || bm == 192 || bm == 256, "sp_gemm_set_route: bm %d", bm);
  SP_REQUIRE(bn == 0 || bn == 256 || bn == 320 || bn == 192, "sp_gemm_set_route: bn %d", bn);
  g_route = route; g_route_bm = bm; g_route_bn = bn;
  return SP_OK;
}

// D[m][n] (fp32, row pitch n) = A[m][k] . W[n][k]^T: the raw fp32 sums, never rounded to fp16.  The 256 x 256 ping-pong
// kernel in its K-slice form with ONE slice (the accumulators leave as 16-byte fp32 stores; no bias, no epilogue).
// Used where the product feeds a softmax whose logits must not pass through fp16: the temporal VAE's single 512-wide
// attention head (the reference runs that VAE in fp32, /root/reference/scripts/generate_video_demo.py:171-175).
extern "C" int sp_gemm_f32out_f16(const void *a_, int64_t lda, const void *w, void *d, int m, int n, int k,
                                  const void *zero_page, void *stream) {
  SP_REQUIRE(a_ && w && d && zero_page, "sp_gemm_f32out_f16: null a/w/d/zero_page");
  SP_REQUIRE(m > 0 && n > 0 && n % 256 == 0, "sp_gemm_f32out_f16: n=%d must be a positive multiple of 256 (m=%d)", n, m);
  SP_REQUIRE(k >= 64 && k % 64 == 0, "sp_gemm_f32out_f16: k=%d must be a multiple of 64", k);
  SP_REQUIRE(lda >= k && lda % 8 == 0, "sp_gemm_f32out_f16: lda=%lld invalid", (long long)lda);
  SP_REQUIRE(((uintptr_t)d & 15) == 0, "sp_gemm_f32out_f16: d must be 16-byte aligned");
  GemmArgs a{};
  a.a = (const f16 *)a_; a.w = (const f16 *)w; a.zero = (const char *)zero_page;
  a.lda = lda; a.mode = SP_A_LINEAR; a.cin = k; a.taps = 1; a.m = m; a.n = n; a.k = k;
  a.oscale = 1.0f; a.bias2_rows = m; a.ldb2 = n;
  a.ksplit = 1; a.partial = (float *)d;
  return spgemm::launch_pp(a, 256, 256, (hipStream_t)stream);
}

extern "C" int sp_gemm_f16(const sp_gemm_desc *d, void *stream) {
  SP_REQUIRE(d != nullptr, "sp_gemm_f16: null descriptor");
  SP_REQUIRE(d->a && d->w && d->d && d->zero_page, "sp_gemm_f16: null a/w/d/zero_page");
  SP_REQUIRE(d->m > 0 && d->n > 0, "sp_gemm_f16: m,n must be positive (m=%d n=%d)", d->m, d->n);
  SP_REQUIRE(d->cin > 0 && d->cin % 64 == 0, "sp_gemm_f16: cin=%d must be a multiple of 64", d->cin);
  SP_REQUIRE(d->n % 64 == 0, "sp_gemm_f16: n=%d must be a multiple of 64", d->n);
  SP_REQUIRE(d->mode >= SP_A_LINEAR && d->mode <= SP_A_TEMPORAL3, "sp_gemm_f16: bad mode %d", d->mode);
  SP_REQUIRE(d->lda >= d->cin && d->lda % 8 == 0, "sp_gemm_f16: lda=%lld invalid", (long long)d->lda);
  SP_REQUIRE(d->ldd % 8 == 0 || d->n_store > 0, "sp_gemm_f16: ldd must be a multiple of 8");
  GemmArgs a{};
#ifdef SP_GEMM_EXPERIMENTS
  { const char *e = getenv("SP_GEMM_DBG"); a.dbg = e ? atoi(e) : 0; }   // ablation builds only (make exp)
  { const char *e = getenv("SP_GEMM_STAGGER"); a.stagger = e ? atoi(e) : 0; }
#endif
  a.a = (const f16 *)d->a; a.w = (const f16 *)d->w; a.bias = d->bias; a.bias2 = d->bias2;
  a.res1 = (const f16 *)d->res1; a.res2 = (const f16 *)d->res2; a.d = (f16 *)d->d;
  a.ln_stats = d->ln_stats; a.ln_colsum = d->ln_colsum;
  a.ln_out = d->ln_out; a.ln_out_eps = d->ln_out_eps;
  a.gn_part = nullptr;
  if (d->gn_part) {
    SP_REQUIRE(!d->geglu && !d->ln_stats && !d->ln_out && !d->euler_out && d->n_store == 0 &&
                   d->m % 256 == 0 && (d->n % 256 == 0 || d->n % 320 == 0) && ((uintptr_t)d->gn_part & 15) == 0,
               "sp_gemm_f16: gn_part needs whole 256-row tiles (m = %d), n a multiple of 256 or 320 (n = %d), no geglu / "
               "folded LayerNorm / ln_out / n_store / Euler tail, and a 16-byte aligned buffer", d->m, d->n);
    a.gn_part = d->gn_part;
  }
  a.w_group_rows = 0; a.w_group_stride = 0;
  if (d->w_group_rows != 0) {
    // (a folded LayerNorm's column sums belong to ONE weight matrix: with one matrix per row group the mean term would be
    // silently wrong.  A forced route is ignored for grouped weights: only the ping-pong tiles pick a matrix per tile.)
    SP_REQUIRE(d->w_group_rows > 0 && d->w_group_rows % 128 == 0 && d->w_group_stride > 0 && d->w_group_stride % 8 == 0 &&
                   d->mode == SP_A_LINEAR && !d->geglu && !d->euler_out && (d->n % 256 == 0 || d->n % 320 == 0) &&
                   !d->ln_stats && !d->ln_colsum,
               "sp_gemm_f16: per-row-group weights need SP_A_LINEAR, no geglu / Euler tail / folded LayerNorm (ln_stats), n a "
               "multiple of 256 or 320, w_group_rows a positive multiple of 128 (got %lld) and w_group_stride a positive multiple "
               "of 8 (got %lld)", (long long)d->w_group_rows, (long long)d->w_group_stride);
    a.w_group_rows = d->w_group_rows; a.w_group_stride = d->w_group_stride;
  }
  if (d->ln_out) {
    const int lbn = d->n % 320 == 0 ? 320 : 256, ltiles = d->n / lbn;       // column tiles per row
    SP_REQUIRE(d->n % lbn == 0 && ltiles >= 1 && ltiles <= 4 && !d->geglu && d->n_store == 0 && !d->euler_out &&
                   d->ln_out_eps > 0.f,
               "sp_gemm_f16: ln_out needs rows of one to four 256- or 320-column tiles (n = 256 ... 1280, got %d), no geglu / "
               "n_store / Euler tail", d->n);
    if (ltiles > 1) {                          // several tiles per row: per-tile sums go through the caller's workspace
      SP_REQUIRE(d->workspace && ((uintptr_t)d->workspace & 7) == 0 &&
                     d->workspace_bytes >= (size_t)d->m * ltiles * 2 * sizeof(float),
                 "sp_gemm_f16: ln_out with n = %d needs a workspace of m * %d bytes (8-byte aligned)", d->n, ltiles * 8);
      a.ln_part = (float *)d->workspace;
    }
  }
  if (d->euler_out) {
    SP_REQUIRE(d->euler_latent && d->n == 64 && d->n_store == 4 && !d->geglu && !d->res1 && !d->res2 && d->oscale == 1.0f,
               "sp_gemm_f16: the Euler tail belongs to conv_out (n = 64, n_store = 4, no residuals, oscale 1)");
    SP_REQUIRE(d->euler_frames > 0 && d->euler_hw > 0 && d->m % (d->euler_frames * d->euler_hw) == 0 && d->euler_sigma > 0.f,
               "sp_gemm_f16: Euler tail: m=%d frames=%d hw=%lld sigma=%g", d->m, d->euler_frames, (long long)d->euler_hw,
               (double)d->euler_sigma);
    if (d->euler_eps_uncond) SP_REQUIRE(d->euler_guidance && d->euler_ld_eps >= 4 && d->euler_ld_eps % 4 == 0,
                                        "sp_gemm_f16: Euler tail: guidance needs euler_guidance and euler_ld_eps");
    const float s2 = d->euler_sigma * d->euler_sigma + 1.0f;
    a.eul_lat = (const f16 *)d->euler_latent; a.eul_out = (f16 *)d->euler_out;
    a.eul_u = (const f16 *)d->euler_eps_uncond; a.eul_gs = d->euler_guidance; a.eul_ldu = d->euler_ld_eps;
    a.eul_frames = d->euler_frames; a.eul_hw = d->euler_hw;
    a.eul_c_out = -d->euler_sigma / sqrtf(s2); a.eul_c_skip = 1.0f / s2;
    a.eul_inv_sigma = 1.0f / d->euler_sigma; a.eul_dt = d->euler_sigma_next - d->euler_sigma;
  }
  if (d->ln_stats)
    SP_REQUIRE(d->ln_colsum && d->mode == SP_A_LINEAR && !d->bias2,
               "sp_gemm_f16: a folded LayerNorm needs ln_colsum, SP_A_LINEAR and no bias2");
  a.zero = (const char *)d->zero_page;
  a.lda = d->lda; a.ldr1 = d->ldr1; a.ldr2 = d->ldr2; a.ldd = d->ldd;
  a.mode = d->mode; a.cin = d->cin;
  a.taps = d->mode == SP_A_CONV3X3 ? 9 : d->mode == SP_A_TEMPORAL3 ? 3 : 1;
  a.m = d->m; a.n = d->n; a.k = a.taps * d->cin;
  a.a2 = nullptr; a.lda2 = 0; a.cin2 = 0;
  if (d->a2) {
    SP_REQUIRE(d->cin2 > 0 && d->cin2 % 64 == 0 && d->lda2 >= d->cin2 && d->lda2 % 8 == 0,
               "sp_gemm_f16: a2 needs cin2 (%d) a positive multiple of 64 and lda2 (%lld) >= cin2, a multiple of 8", d->cin2,
               (long long)d->lda2);
    SP_REQUIRE(!d->geglu && !d->ln_stats && !d->ln_out && !d->euler_out && d->n_store == 0 && d->w_group_rows == 0 &&
                   (d->n % 256 == 0 || d->n % 320 == 0) && !(d->gn_part && (d->res1 || d->res2)),
               "sp_gemm_f16: a2 (extra linear tap) runs on the 256-row ping-pong tiles only: n a multiple of 256 or 320 (n = %d), "
               "no geglu / folded LayerNorm / ln_out / n_store / Euler tail / per-group weights, gn_part only without residuals",
               d->n);
    a.a2 = (const f16 *)d->a2; a.lda2 = d->lda2; a.cin2 = d->cin2;
    a.k += d->cin2;
  }
  a.oscale = d->oscale; a.r1scale = d->r1scale; a.r2scale = d->r2scale;
  a.geglu = d->geglu; a.n_store = d->n_store;
  a.bias2_rows = d->bias2_rows > 0 ? d->bias2_rows : d->m;
  a.ldb2 = d->ldb2 > 0 ? d->ldb2 : d->n;
  if (d->res1) SP_REQUIRE(d->ldr1 % 8 == 0, "sp_gemm_f16: ldr1 must be a multiple of 8");
  if (d->res2) SP_REQUIRE(d->ldr2 % 8 == 0, "sp_gemm_f16: ldr2 must be a multiple of 8");
  if (d->mode == SP_A_CONV3X3) {
    SP_REQUIRE(d->stride == 1 || d->stride == 2, "sp_gemm_f16: stride must be 1 or 2");
    SP_REQUIRE(d->n_img > 0 && d->hin > 0 && d->win > 0 && d->hout > 0 && d->wout > 0,
               "sp_gemm_f16: conv geometry must be positive");
    const int hv = d->hin << (d->upsample2x ? 1 : 0), wv = d->win << (d->upsample2x ? 1 : 0);
    SP_REQUIRE(d->hout == (hv + 2 - 3) / d->stride + 1 && d->wout == (wv + 2 - 3) / d->stride + 1,
               "sp_gemm_f16: conv output %dx%d inconsistent with input %dx%d stride %d", d->hout,
               d->wout, hv, wv, d->stride);
    SP_REQUIRE((int64_t)d->n_img * d->hout * d->wout == d->m, "sp_gemm_f16: m != n_img*hout*wout");
    a.n_img = d->n_img; a.hin = d->hin; a.win = d->win; a.hout = d->hout; a.wout = d->wout;
    a.stride = d->stride; a.ups = d->upsample2x ? 1 : 0;
  } else if (d->mode == SP_A_TEMPORAL3) {
    SP_REQUIRE(d->frames > 0 && d->hw > 0 && d->m % (d->frames * d->hw) == 0,
               "sp_gemm_f16: temporal geometry: m=%d frames=%d hw=%lld", d->m, d->frames,
               (long long)d->hw);
    a.frames = d->frames; a.hw = d->hw;
  }
  if (d->geglu) SP_REQUIRE(d->n % 128 == 0, "sp_gemm_f16: geglu needs n %% 128 == 0");
  hipStream_t s = (hipStream_t)stream;
  return dispatch(a, d, s);
}

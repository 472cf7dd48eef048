// Large-tile implicit-GEMM for gfx950: 256 x BN x 32 per workgroup (BN = 256 or 320), 8 waves as 2(M) x 4(N),
// 4-deep LDS ring, ping-pong schedule.  Same contract and epilogue as gemm.hip (see there for the operand
// gather and the swizzle idea); this variant exists because the per-CU L2->LDS path (~64 B/clk) feeds a
// 256x128 tile only ~1.0 PFLOP/s worth of operands: a 256x256 (256x320) tile needs half (0.44x) the bytes per
// FLOP, and every N of the SVD UNet is a multiple of 256 or 320.
//
// Schedule.  One K-step is 32 channels: 32 KB (36 KB) of operands, 32 (40) MFMAs per wave.  Each K-step of a
// wave is a READ phase (issue the LDS-DMA for K-step k+3, ds_read the 12 (13) operand fragments of K-step k)
// followed by a COMPUTE phase (the MFMAs, operands in registers), with a workgroup barrier after each phase.
// Waves 4-7 run one phase behind waves 0-3 (they share SIMDs pairwise: w and w+4), so on every SIMD one wave is
// in its MFMA phase while the other fetches operands.  LDS-DMA stays in flight across barriers behind counted
// s_waitcnt vmcnt (K-steps k+2 and k+3 outstanding while k+1 is retired one phase before its first read).
//
// LDS image per stage: rows of 64 B (32 halves), 16-byte chunk c of row r stored at chunk c ^ S[(r>>2)&3],
// S = {0,2,3,1}: makes the ds_read_b128 of a 16x16x32 operand (16 rows x one chunk per 16-lane group)
// bank-conflict free; applied on the DMA *source* address, undone in the read address.
#include "gemm_args.h"
#include <stdlib.h>
#include <type_traits>

namespace spgemm {
namespace {

constexpr int PBK = 32, PNW = 8;


#ifdef SP_GEMM_EXPERIMENTS
// per-workgroup phase timestamps (100 MHz wall clock) + hardware id, read back by tools/pp_trace.py
constexpr int PP_TRACE_WGS = 16384, PP_TRACE_SLOTS = 12;
__device__ long long g_pp_trace[PP_TRACE_WGS * PP_TRACE_SLOTS];
#define PP_TRACE(slot)                                                                              \
  do {                                                                                              \
    if (threadIdx.x == 0 && blockIdx.x < PP_TRACE_WGS)                                              \
      g_pp_trace[blockIdx.x * PP_TRACE_SLOTS + (slot)] = wall_clock64();                            \
  } while (0)
// shader-clock stamp (s_memtime): with the 100 MHz stamps it gives the clock the K loop really ran at
#define PP_TRACE_CLK(slot)                                                                          \
  do {                                                                                              \
    if (threadIdx.x == 0 && blockIdx.x < PP_TRACE_WGS)                                              \
      g_pp_trace[blockIdx.x * PP_TRACE_SLOTS + (slot)] = (long long)__builtin_amdgcn_s_memtime();   \
  } while (0)
#else
#define PP_TRACE(slot) do {} while (0)
#define PP_TRACE_CLK(slot) do {} while (0)
#endif

__device__ __forceinline__ int swz4(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

#ifndef SP_PP_STAGES_192_256
#define SP_PP_STAGES_192_256 4
#endif
constexpr int pp_stages(int bm, int bn) { return bm == 128 ? 3 : (bm == 192 && bn == 256 ? SP_PP_STAGES_192_256 : 4); }

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// waits until at most `ksteps_left` K-steps of this wave's DMA (L instructions each) are outstanding
template <int L>
__device__ __forceinline__ void wait_dma(int ksteps_left) {
  if (ksteps_left >= 3) wait_vm<3 * L>();
  else if (ksteps_left == 2) wait_vm<2 * L>();
  else if (ksteps_left == 1) wait_vm<L>();
  else wait_vm<0>();
}

// Shared epilogue of the ping-pong kernels.  Bias and time-embedding row are already in the accumulators (kernel
// prologue); here every wave applies output scale / GEGLU, converts to fp16 and writes its part of the tile into
// LDS (the idle ring; 160 KB for 256x320), one barrier, then the whole tile leaves as 16-byte coalesced stores.
// LDS image: rows of bno halves, unpadded, 16-byte chunk c of row r at chunk c ^ (r & 7) (keeps the 8-byte fragment
// writes and the 16-byte row reads at <= 2-way bank conflicts).  The residual reads are issued before/while staging
// so that their latency overlaps the LDS round trip, and all arithmetic precedes the first store: on gfx9 stores
// count in vmcnt like loads, so a load consumed after a store was issued waits for that store's acknowledgement.
#define WROWS_OF(pbm, wtm) ((pbm) / (wtm))      /* wave rows of a tile */
template <int PBM, int BN, int TN, int TM, int WTN, int WTM, bool GEGLU, int NT, bool GNRES = false>
__device__ __forceinline__ void pp_epilogue(const GemmArgs &p, f32x4 (&acc)[TN][TM], char *smem, int tile_m,
                                            int tile_n, int wm, int wn, int tid, int fr, int fq) {
  constexpr int bno = GEGLU ? BN / 2 : BN;
  constexpr int cpr = bno >> 3;                          // 16-byte chunks per row (a multiple of 8)
  constexpr int TNO = GEGLU ? TN / 2 : TN;
  constexpr int NCH = PBM * cpr, ITERS = (NCH + NT - 1) / NT, ITERS_A = ITERS / 2;
  static_assert(cpr % 8 == 0, "chunk swizzle works on aligned groups of 8 chunks");
  const int ncols_total = GEGLU ? p.n / 2 : p.n;
  const int nstore = p.n_store > 0 ? p.n_store : ncols_total;
  const int64_t mbase = (int64_t)tile_m * PBM;

  // chunk of this thread in copy-out iteration `it`; chunks that will not be stored read the zero page, which keeps
  // the prefetch straight-line (loads under divergent branches make the compiler drain vmcnt before each one)
  auto res_src = [&](const f16 *res, int64_t ldr, int it) -> const f16 * {
    const int idx = tid + it * NT;
    const int r = idx / cpr, c = idx - r * cpr;
    const int64_t m = mbase + r;
    const int col = tile_n * bno + c * 8;
    const bool ok = idx < NCH && m < p.m && col + 8 <= nstore;
    return ok ? res + m * ldr + col : (const f16 *)p.zero;
  };

  // ---- folded LayerNorm: acc <- rstd[m]*(acc - mean[m]*colsum[n]) + bias[n] (the accumulators were started at zero)
  if (p.ln_stats) {
    f32x2 st[TM];
    const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.ln_stats, 0, (int)min((int64_t)p.m * 8, (int64_t)0x7fffffff), 0x00020000);
#pragma unroll
    for (int j = 0; j < TM; ++j)       // (rows past M fall outside the buffer and read zeros; they are never stored)
      st[j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(
          s_rsrc, (int)((mbase + wm * WTM + j * 16 + fr) * 8), 0, 0));
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = tile_n * BN + wn * WTN + i * 16 + 4 * fq;
      const f32x4 cs = *(const f32x4 *)(p.ln_colsum + n);
      f32x4 b = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) b = *(const f32x4 *)(p.bias + n);
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = (acc[i][j] - st[j][0] * cs) * st[j][1] + b;
      __builtin_amdgcn_sched_barrier(0);      // one column sub-tile at a time (hoisting all row sums costs 40 registers)
    }
  }
  // ---- GroupNorm statistics of the NEXT norm (gn_part; host: no residuals / geglu / LayerNorm fold, m a multiple of the
  // tile height): per output column, (sum, sum of squares) of this wave's WTM rows, from the fp32 accumulators (bias and
  // time-embedding row are in them; nothing is added later).  Eight in-register adds per column, then the 16 lanes that hold
  // the same columns of different rows are folded with four DPP steps (a fixed butterfly: deterministic).  One 32-byte
  // record per lane quad of columns; a small kernel folds tiles and columns into (mean, rstd) per (instance, group).
  if constexpr (!GEGLU) {
    if (p.gn_part && !p.res1 && !p.res2) {
      auto row16_sum = [](float v) {
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
        v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
        return v;
      };
      float *dst = p.gn_part + (((int64_t)tile_m * WROWS_OF(PBM, WTM) + wm) * p.n + (int64_t)tile_n * BN + wn * WTN) * 2;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        f32x4 sm = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const f32x4 v = acc[i][j] * p.oscale;
          sm += v;
          sq += v * v;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { sm[r] = row16_sum(sm[r]); sq[r] = row16_sum(sq[r]); }
        if (fr == 0) {
          float *o = dst + (i * 16 + 4 * fq) * 2;
          *(f32x4 *)o = (f32x4){sm[0], sq[0], sm[1], sq[1]};
          *(f32x4 *)(o + 4) = (f32x4){sm[2], sq[2], sm[3], sq[3]};
        }
      }
    }
  }
  f16x8 q1[ITERS];
  if (p.res1) {
#pragma unroll
    for (int it = 0; it < ITERS_A; ++it) q1[it] = *(const f16x8 *)res_src(p.res1, p.ldr1, it);
  }
  PP_TRACE(4);

  // ---- stage the tile
#pragma unroll
  for (int i = 0; i < TNO; ++i) {
    const int col = (GEGLU ? (wn * WTN) / 2 : wn * WTN) + i * 16 + 4 * fq;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      f16x4 o;
      if constexpr (!GEGLU) {
        const f32x4 v = acc[i][j] * p.oscale;
        o = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
      } else {
        const f32x4 h = acc[2 * i][j] * p.oscale, g = acc[2 * i + 1][j];
        const f32x2 ga = gelu2_f((f32x2){g[0], g[1]}), gb = gelu2_f((f32x2){g[2], g[3]});
        o = (f16x4){(f16)(h[0] * ga[0]), (f16)(h[1] * ga[1]), (f16)(h[2] * gb[0]), (f16)(h[3] * gb[1])};
      }
      const int row = wm * WTM + j * 16 + fr;
      *(f16x4 *)(smem + row * (bno * 2) + ((((col >> 3) ^ (fr & 7))) << 4) + (col & 4) * 2) = o;
    }
  }
  if (p.res1) {
#pragma unroll
    for (int it = ITERS_A; it < ITERS; ++it) q1[it] = *(const f16x8 *)res_src(p.res1, p.ldr1, it);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  PP_TRACE(6);

  // ---- copy out
  f16x8 o[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = tid + it * NT;
    const int r = idx / cpr, c = idx - r * cpr;      // (idx >= NCH reads idle ring memory, never stored)
    o[it] = *(const f16x8 *)(smem + r * (bno * 2) + ((c ^ (r & 7)) << 4));
  }
  // (GroupNorm column sums with residuals, below: the final values go back into this LDS image while they are stored)
  // (its own kernel instantiation, EXP = 2048: the default one keeps its register allocation)
  const bool gn_res = GNRES && !GEGLU && PBM == 256 && p.gn_part && (p.res1 || p.res2);
  if (gn_res) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave has read its part of the tile
  if (p.res1 || p.res2) {
    constexpr int G = (ITERS + 3) / 4;   // res2 (one GEMM family) is fetched here, in groups to bound registers
#pragma unroll
    for (int g0 = 0; g0 < ITERS; g0 += G) {
      f16x8 q2[G];
      if (p.res2) {
#pragma unroll
        for (int it = g0; it < g0 + G && it < ITERS; ++it) q2[it - g0] = *(const f16x8 *)res_src(p.res2, p.ldr2, it);
      }
#pragma unroll
      for (int it = g0; it < g0 + G && it < ITERS; ++it) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (float)o[it][e];
        if (p.res1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += p.r1scale * (float)q1[it][e];
        }
        if (p.res2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += p.r2scale * (float)q2[it - g0][e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[it][e] = (f16)f[e];
      }
    }
  }
  PP_TRACE(7);
  // ---- LayerNorm statistics of the rows about to be stored (ln_out; host: a row is one tile, or two with ln_part): every
  // thread leaves (sum, sum of squares) of its eight final fp16 values in LDS -- the staged tile is in registers by now
  // -- and one thread per row folds the row's chunks in a fixed order.  fp32 sums over 256 / 320 values.
  if constexpr (!GEGLU) {
    if (p.ln_out) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave has read its part of the tile
      float2 *part = (float2 *)smem;                                       // [PBM][cpr]
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * NT;
        if (idx < NCH) {
          float sm = 0.f, sq = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float f = (float)o[it][e]; sm += f; sq = fmaf(f, f, sq); }
          part[idx] = make_float2(sm, sq);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (tid < PBM && mbase + tid < p.m) {
        float sm = 0.f, sq = 0.f;
        for (int c = 0; c < cpr; ++c) { const float2 v = part[tid * cpr + c]; sm += v.x; sq += v.y; }
        if (p.ln_part) {                       // the row continues in the next tile: raw sums, folded by ln_part_finalize
          *(float2 *)(p.ln_part + ((mbase + tid) * p.tiles_n + tile_n) * 2) = make_float2(sm, sq);
        } else {
          const float inv = 1.0f / (float)bno, mean = sm * inv;
          float var = sq * inv - mean * mean;
          if (var < 0.f) var = 0.f;
          *(float2 *)(p.ln_out + (mbase + tid) * 2) = make_float2(mean, rsqrtf(var + p.ln_out_eps));
        }
      }
    }
  }
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = tid + it * NT;
    const int r = idx / cpr, c = idx - r * cpr;
    const int64_t m = mbase + r;
    const int col = tile_n * bno + c * 8;
    if (idx >= NCH || m >= p.m || col >= nstore) continue;
    if (col + 8 <= nstore) {
#ifdef SP_GEMM_EXPERIMENTS
      // (non-temporal stores measured here too: no difference on any shape -- a one-tile workgroup ends behind its stores,
      // nothing waits for their acknowledgement; profiles/r04_nt_stores_ab.txt)
      if (p.dbg & 512) __builtin_nontemporal_store(o[it], (f16x8 *)(p.d + m * p.ldd + col));
      else
#endif
      *(f16x8 *)(p.d + m * p.ldd + col) = o[it];
      if (gn_res) *(f16x8 *)(smem + r * (bno * 2) + ((c ^ (r & 7)) << 4)) = o[it];
    } else {   // ragged last chunk (n_store): the residuals were not prefetched for it
      for (int e = 0; e < nstore - col; ++e) {
        float f = (float)o[it][e];
        if (p.res1) f += p.r1scale * (float)p.res1[m * p.ldr1 + col + e];
        if (p.res2) f += p.r2scale * (float)p.res2[m * p.ldr2 + col + e];
        p.d[m * p.ldd + col + e] = (f16)f;
      }
    }
  }
  // ---- GroupNorm statistics of the NEXT norm when residuals were added (gn_part; host: m a multiple of PBM = 256, no
  // n_store): the sums have to come from the FINAL values, which exist in the copy-out layout only (8 channels of one
  // row per thread, the column changing from pass to pass).  So the final fp16 tile goes back into the idle LDS image
  // while it is stored (LDS traffic does not queue behind the stores), then every thread of the first 2*RG*Q sums one quad of
  // columns over its share of a 128-row half, the RG partials of a (half, quad) meet in LDS and are added in a fixed order.
  // Same record format as the accumulator path above ([tile][half][column][2]); sums of the rounded values.
  if constexpr (GNRES && !GEGLU && PBM == 256) {
    if (gn_res) {
      constexpr int Q = bno / 4, RG = NT / (2 * Q), RPG = (128 + RG - 1) / RG;     // quads per row, row groups per half, rows each
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");            // the final tile is in LDS
      const int quad = tid % Q, grp = tid / Q;
      float sm[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
      if (grp < 2 * RG) {
        const int half = grp / RG, sub = grp - half * RG;
        const int r0 = half * 128 + sub * RPG, r1 = min(r0 + RPG, half * 128 + 128);
        const int col = quad * 4;
        // eight independent LDS reads in flight per pass (a one-read-at-a-time loop is 43 exposed round trips); rows past
        // the group's end re-read its last row with weight zero
        for (int k = 0; k < RPG; k += 8) {
          f16x4 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int r = min(r0 + k + u, r1 - 1);
            v[u] = *(const f16x4 *)(smem + r * (bno * 2) + ((((col >> 3) ^ (r & 7))) << 4) + (col & 4) * 2);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float wgt = r0 + k + u < r1 ? 1.0f : 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float f = (float)v[u][e] * wgt; sm[e] += f; sq[e] = fmaf(f, f, sq[e]); }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");            // the tile has been read: its memory takes the partials
      float *pt = (float *)smem;
      if (grp < 2 * RG) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { pt[(grp * Q + quad) * 8 + 2 * e] = sm[e]; pt[(grp * Q + quad) * 8 + 2 * e + 1] = sq[e]; }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (tid < 2 * Q) {
        const int half = tid / Q, qd = tid - half * Q;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        for (int sub = 0; sub < RG; ++sub) {
          const float *src = pt + ((half * RG + sub) * Q + qd) * 8;
          a += *(const f32x4 *)src; b += *(const f32x4 *)(src + 4);
        }
        float *dst = p.gn_part + (((int64_t)tile_m * 2 + half) * p.n + (int64_t)tile_n * bno + qd * 4) * 2;
        *(f32x4 *)dst = a;
        *(f32x4 *)(dst + 4) = b;
      }
    }
  }
}

#ifdef SP_GEMM_EXPERIMENTS
// COMPUTE phase of one K-step with this wave's LDS-DMA pieces for K-step kt+3 spread between its MFMAs (an LDS-DMA
// piece costs the issuing wave ~60 cycles among MFMAs but 100-185 in a phase that also carries ds_reads).  One
// MFMA stream for every wave; `pieces` is a wave-uniform bit mask (bit pc: issue piece pc; A pieces first).
template <int TN, int TM, int NAP, int NBP, int NWV>
__device__ __forceinline__ void mfma_block(f32x4 (&acc)[TN][TM], const f16x8 (&fw)[TN], const f16x8 (&fa)[TM],
                                           const f16 *(&aptr)[NAP], int (&astep)[NAP], const f16 *(&bptr)[NBP],
                                           char *dsa, char *dsb, int pieces) {
  constexpr int NP = NAP + NBP, NM = TN * TM, GAP = NM / (NP + 1);
#pragma unroll
  for (int i = 0; i < TN; ++i) {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
      const int q1 = i * TM + j + 1;
      if (q1 % GAP == 0 && q1 / GAP <= NP) {
        const int pc = q1 / GAP - 1;
        if (pieces & (1 << pc)) {
          if (pc < NAP) {
            glds16(aptr[pc], dsa + pc * (NWV * 1024));
            aptr[pc] += astep[pc];
          } else {
            glds16(bptr[pc - NAP], dsb + (pc - NAP) * (NWV * 1024));
            bptr[pc - NAP] += PBK;
          }
        }
      }
    }
  }
}
#endif

// BM = 256: one workgroup of 8 waves per CU (4-deep ring).  BM = 192: same, for row counts where 256-row tiles leave a
// third of the CUs idle in the only round (8064 rows x 1280 columns).  BM = 128: FOUR waves (1 x 4, the same 128 x BN/4
// tile per wave as BM = 256), 3-deep ring, so that TWO workgroups share a CU: each SIMD holds one wave of either, and
// one workgroup's prologue/epilogue runs under the other's K loop (short-K GEMMs).
template <int BM, int BN, int EXP>
__global__ __launch_bounds__(BM == 128 ? 256 : 512, 2) void gemm_pp_kernel(const GemmArgs p) {
  constexpr int PBM = BM;
  constexpr bool SPLITK = (EXP & 128) != 0;       // K-slice variant (only 256 x 256 is instantiated), see below
  // Extra linear tap (EXP & 4096, its own instantiations): behind the taps of the convolution the K loop runs over the
  // channels of a SECOND tensor's rows (same row index as the output) against the weight columns that follow -- a resnet's
  // 1x1 shortcut convolution folded into its second 3x3 convolution: the skip tensor is neither written nor read
  constexpr bool A2 = (EXP & 4096) != 0;
  constexpr int NWV = BM == 128 ? 4 : 8, NT = NWV * 64, WROWS = NWV / 4;   // waves, threads, wave rows (x 4 columns)
  constexpr int PSTAGES = pp_stages(BM, BN), PDIST = PSTAGES - 1;
  constexpr int TN = BN / 4 / 16;                 // weight sub-tiles per wave (4 or 5)
  constexpr int TM = BM / WROWS / 16;             // activation sub-tiles per wave
  constexpr int WTN = BN / 4, WTM = BM / WROWS;
  constexpr int A_BYTES = PBM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
  // 1-KiB DMA pieces (16 rows x 64 B) are dealt to waves as piece = j*NWV + wave; when the count is not a
  // multiple of NWV it is 4 mod 8, so waves 0-3 ("early" half) issue one piece more than waves 4-7.
  constexpr int A_PIECES = BM / 16, B_PIECES = BN / 16;
  constexpr int A_LOADS = (A_PIECES + NWV - 1) / NWV, A_LOADS_HI = A_PIECES / NWV;
  constexpr int A_SPLIT = A_PIECES % NWV == 0 ? NWV : A_PIECES % NWV;
  constexpr int B_LOADS_LO = (B_PIECES + NWV - 1) / NWV;  // waves 0..(B_PIECES%NWV - 1) (all waves if divisible)
  constexpr int B_LOADS_HI = B_PIECES / NWV;              // the other waves
  constexpr int B_SPLIT = B_PIECES % NWV == 0 ? NWV : B_PIECES % NWV;   // waves below this index take B_LOADS_LO
  static_assert((B_SPLIT == NWV || B_SPLIT == 4) && (A_SPLIT == NWV || A_SPLIT == 4), "wave halves must have uniform DMA counts");
#ifdef SP_GEMM_EXPERIMENTS
  // (EXP & 16, timing only, results are wrong: activation pieces are issued for tap 0 only -- the DMA-instruction
  // count of a kernel that keeps a halo tile of the activations in LDS; the waits count weight pieces only)
  constexpr bool HALO_COUNT = (EXP & 16) != 0;
  constexpr int L_EARLY = (HALO_COUNT ? 0 : A_LOADS) + B_LOADS_LO;
  constexpr int L_LATE = (HALO_COUNT ? 0 : (A_SPLIT == NWV ? A_LOADS : A_LOADS_HI)) + (B_SPLIT == NWV ? B_LOADS_LO : B_LOADS_HI);
#else
  constexpr int L_EARLY = A_LOADS + B_LOADS_LO;   // waves 0-3
  constexpr int L_LATE = (A_SPLIT == NWV ? A_LOADS : A_LOADS_HI) + (B_SPLIT == NWV ? B_LOADS_LO : B_LOADS_HI);   // waves 4-7
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const bool late = NWV == 8 && wave >= 4;
#ifdef SP_GEMM_EXPERIMENTS
  // lock-step experiment: the first round of workgroups (one per CU) starts in four phase groups, so that later
  // rounds keep prologue reads / K loops / epilogue stores of neighbouring CUs apart in time
  if (p.stagger > 0 && blockIdx.x < 256) {
    const int d = ((blockIdx.x >> 3) & 3) * p.stagger;
    for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(30);   // 30 * 64 cycles ~ 1 us at 1.9 GHz
  }
#endif
  PP_TRACE(0);
#ifdef SP_GEMM_EXPERIMENTS
  if (tid == 0 && blockIdx.x < PP_TRACE_WGS) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
    g_pp_trace[blockIdx.x * PP_TRACE_SLOTS + 5] = ((long long)xcc << 32) | hw;
  }
#endif

  constexpr int GM = BM == 128 ? 8 : 4;
  const int nwg = p.tiles_m * p.tiles_n;
  // split-K (few-row contractions with a long K, SPLITK instances): workgroup = (tile, K slice); slice s sums K-steps
  // [s*nk/S, (s+1)*nk/S) into an fp32 slab of the workspace, splitk_reduce_kernel (gemm.hip) applies the epilogue
  int bid = blockIdx.x, kslice = 0;
  if constexpr (SPLITK) { kslice = bid / nwg; bid -= kslice * nwg; }
  const int t = xcd_remap(bid, nwg);
  const int per_group = GM * p.tiles_n;
  const int group = t / per_group;
  const int first_m = group * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int in_group = t - group * per_group;
  const int tile_n = in_group / gsz;
  const int tile_m = first_m + (in_group - tile_n * gsz);

  // ---------------------------------------------------------------- per-lane gather state
  const int lrow = lane >> 2, lchunk = lane & 3;  // 16 rows x 4 chunks per 1-KiB piece
  const int cpt = p.cin >> 5;                     // K-steps per tap

  int a_i0[A_LOADS], a_i1[A_LOADS], a_i2[A_LOADS];
  bool a_in[A_LOADS];
  int schunk_a[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int r = (i * NWV + wave) * 16 + lrow;
    const int m = tile_m * PBM + r;
    a_in[i] = m < p.m && r < PBM;
    schunk_a[i] = (lchunk ^ swz4(r)) * 8;
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
      if constexpr (A2) {
        if (tap == p.taps) {                      // the extra tap: row m of a2 (wave-uniform branch)
          const int r = (i * NWV + wave) * 16 + lrow;
          if (a_in[i]) {
            aptr[i] = p.a2 + ((int64_t)tile_m * PBM + r) * p.lda2 + schunk_a[i];
            astep[i] = PBK;
          } else {
            aptr[i] = (const f16 *)(p.zero + lchunk * 16);
            astep[i] = 0;
          }
          continue;
        }
      }
#ifdef SP_GEMM_EXPERIMENTS
      // (EXP & 8, timing only: taps > 0 read the zero page = what an LDS-resident halo tile would save the memory pipe)
      if (a_in[i] && !((EXP & 8) && tap > 0)) {
#else
      if (a_in[i]) {
#endif
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
        astep[i] = PBK;
      } else {
        aptr[i] = (const f16 *)(p.zero + lchunk * 16);
        astep[i] = 0;
      }
    }
  };

  // W pieces: piece index = j*8 + wave (j < B_LOADS_LO; the last j only exists for waves < B_SPLIT)
  const f16 *bptr[B_LOADS_LO];
#pragma unroll
  for (int j = 0; j < B_LOADS_LO; ++j) {
    const int piece = j * NWV + wave;
    const int r = piece * 16 + lrow;
    const int n = tile_n * BN + (r < BN ? r : 0);
    bptr[j] = p.w + (int64_t)n * p.k + (lchunk ^ swz4(r)) * 8;
    // per-row-group weights (a GroupNorm folded into this linear layer: one scaled copy per frame; host: no tile straddles)
    if (p.w_group_rows > 0) bptr[j] += ((int64_t)tile_m * PBM / p.w_group_rows) * p.w_group_stride;
  }

  bool skip_a = false;                            // (experiments build: see HALO_COUNT)
  auto stage = [&](int slot) {
    char *sa = smem + slot * STAGE;
    char *sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      if ((i < A_LOADS_HI || wave < A_SPLIT) && !skip_a) {     // wave-uniform
        glds16(aptr[i], sa + (i * NWV + wave) * 1024);
        aptr[i] += astep[i];
      }
    }
#pragma unroll
    for (int j = 0; j < B_LOADS_LO; ++j) {
      if (j < B_LOADS_HI || wave < B_SPLIT) {     // wave-uniform
        glds16(bptr[j], sb + (j * NWV + wave) * 1024);
        bptr[j] += PBK;
      }
    }
  };

  // ---------------------------------------------------------------- main loop
  // Accumulators start at bias (+ the per-image time-embedding row) instead of adding them in the epilogue.  The
  // bias loads are issued ahead of the first LDS-DMA (so the counted vmcnt waits below still mean "K-step landed")
  // and consumed after it; nothing else delays that first DMA: the 128-160 accumulator registers are written
  // while it is in flight.  The time-embedding rows (first conv of a residual block only, K >= 2880) are loaded
  // after the DMA, straight into the accumulators; that tile's first wait then also covers them.
  f32x4 acc[TN][TM], bias_v[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    bias_v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // (a folded LayerNorm applies its bias in the epilogue, after the row scaling)
    if (p.bias && !p.ln_stats && !SPLITK)
      bias_v[i] = *(const f32x4 *)(p.bias + tile_n * BN + wn * WTN + i * 16 + 4 * (lane >> 4));
  }
  __builtin_amdgcn_sched_barrier(0);

  int nk = p.k >> 5, kbeg = 0;
  if constexpr (SPLITK) {
    kbeg = (int)((int64_t)kslice * nk / p.ksplit);
    nk = (int)((int64_t)(kslice + 1) * nk / p.ksplit) - kbeg;
  }
  const int fr = lane & 15, fq = lane >> 4;
  const int rd_chunk = (fq ^ swz4(fr)) << 4;      // fragment rows are (multiple of 16) + fr
  int offw[TN], offa[TM];
#pragma unroll
  for (int i = 0; i < TN; ++i) offw[i] = (wn * WTN + i * 16 + fr) * 64 + rd_chunk;
#pragma unroll
  for (int j = 0; j < TM; ++j) offa[j] = (wm * WTM + j * 16 + fr) * 64 + rd_chunk;

  int staged = 0, in_tap = 0, tap = 0, stage_slot = 0, read_slot = 0;
  auto stage_next = [&]() {
    // (the extra tap, index p.taps, is the last one and may be longer than a regular tap: no further switch behind it)
    if (in_tap == cpt && (!A2 || tap < p.taps)) { ++tap; in_tap = 0; set_tap(tap); }
#ifdef SP_GEMM_EXPERIMENTS
    if constexpr ((EXP & 16) != 0) skip_a = tap > 0;
#endif
    stage(stage_slot);
    stage_slot = stage_slot + 1 == PSTAGES ? 0 : stage_slot + 1;
    ++staged; ++in_tap;
  };
  tap = kbeg / cpt;
  in_tap = kbeg - tap * cpt;
  set_tap(tap);
  if (kbeg) {                                   // a K slice starts in the middle of the weight rows (and of a tap)
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) aptr[i] += in_tap * astep[i];
#pragma unroll
    for (int j = 0; j < B_LOADS_LO; ++j) bptr[j] += (int64_t)kbeg * PBK;
  }
  PP_TRACE(10);
#pragma unroll
  for (int s = 0; s < PDIST; ++s)
    if (s < nk) stage_next();
  PP_TRACE(11);
  __builtin_amdgcn_sched_barrier(0);
  if (p.bias2 && !SPLITK) {
    int brow[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = tile_m * PBM + wm * WTM + j * 16 + fr;
      brow[j] = m < p.m ? m / (int)p.bias2_rows : 0;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = tile_n * BN + wn * WTN + i * 16 + 4 * fq;
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = *(const f32x4 *)(p.bias2 + (int64_t)brow[j] * p.ldb2 + n) + bias_v[i];
      __builtin_amdgcn_sched_barrier(0);      // one row of tiles at a time: 40 address pairs at once would spill
    }
  } else {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = bias_v[i];
  }
  // K-step 0 landed (this wave's part), later ones may stay in flight
  {
    const int left = min(PDIST - 1, nk - 1);
    if (late) wait_dma<L_LATE>(left); else wait_dma<L_EARLY>(left);
  }
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();
  PP_TRACE(1);
  PP_TRACE_CLK(8);

#ifdef SP_GEMM_EXPERIMENTS
  constexpr int dbg = EXP;   // timing experiments only: 1 no DMA in loop, 2 no MFMA, 4 DMA between MFMAs, 64 no ds_read, 32 GroupNorm+SiLU on the A fragments
  constexpr bool DMA_IN_COMPUTE = (dbg & 4) != 0;
#endif
  for (int kt = 0; kt < nk; ++kt) {
    // ---- READ phase: operand fragments first, then the LDS-DMA for K-step kt+3.  The DMA instructions
    // queue behind the CU's single 64 B/clk texture-address path (about 100 cycles each when four waves issue
    // together); placed after the ds_reads they stall a wave that has nothing left to issue anyway.
    const bool issue = kt + PDIST < nk;
    const char *sa = smem + read_slot * STAGE;
    read_slot = read_slot + 1 == PSTAGES ? 0 : read_slot + 1;
    const char *sb = sa + A_BYTES;
    f16x8 fw[TN], fa[TM];
#ifdef SP_GEMM_EXPERIMENTS
    if ((dbg & 64) && kt > 0) {
#pragma unroll
      for (int i = 0; i < TN; ++i) fw[i] = (f16x8){1, 2, 3, 4, 5, 6, 7, (f16)kt};
#pragma unroll
      for (int j = 0; j < TM; ++j) fa[j] = (f16x8){1, 2, 3, 4, 5, 6, 7, (f16)lane};
    } else
#endif
    {
#pragma unroll
      for (int i = 0; i < TN; ++i) fw[i] = *(const f16x8 *)(sb + offw[i]);
#pragma unroll
      for (int j = 0; j < TM; ++j) fa[j] = *(const f16x8 *)(sa + offa[j]);
    }
#ifdef SP_GEMM_EXPERIMENTS
    if constexpr ((dbg & 32) != 0) {
      // timing only: GroupNorm apply + SiLU in the CONSUMER's A path -- y = silu(x * scale + shift)
      // on every activation fragment after its ds_read, in packed fp16 (the cheapest form: 4 packed ops + 4 half-rate
      // transcendentals per pair); scale / shift per channel would come from LDS, here they are lane constants.  Results
      // are wrong (and out-of-image taps would have to stay zero): the arm prices the instruction stream only.
      const f16x2 sc2 = {(f16)(1.0f + 0.001f * lane), (f16)1.0f}, sh2 = {(f16)0.01f, (f16)(0.002f * lane)};
#pragma unroll
      for (int j = 0; j < TM; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f16x2 v = {fa[j][2 * e], fa[j][2 * e + 1]};
          v = __builtin_elementwise_fma(v, sc2, sh2);
          const f16x2 t = __builtin_elementwise_exp2(v * (f16x2){(f16)-1.4427f, (f16)-1.4427f}) + (f16x2){(f16)1.0f, (f16)1.0f};
          v = v * (f16x2){(f16)__builtin_amdgcn_rcph(t[0]), (f16)__builtin_amdgcn_rcph(t[1])};
          fa[j][2 * e] = v[0]; fa[j][2 * e + 1] = v[1];
        }
      }
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifdef SP_GEMM_EXPERIMENTS
    if constexpr (DMA_IN_COMPUTE) {
      if (issue && in_tap == cpt) { ++tap; in_tap = 0; set_tap(tap); }     // addresses for the coming pieces
      // late waves: this wave's pieces of K-step kt+1 must have landed; kt+2 may stay in flight (kt+3 not issued yet)
      int left = min(PDIST - 2, nk - 2 - kt);
      if (left < 0) left = 0;
      if (late) wait_dma<L_LATE>(left);
    } else
#endif
    {
#ifdef SP_GEMM_EXPERIMENTS
      if constexpr (!(dbg & 1))
#endif
      { if (issue) stage_next(); }
      // K-steps issued beyond kt+1 so far: up to kt+PDIST
      int left = min(PDIST - 1, nk - 2 - kt);
      if (left < 0) left = 0;
      if (late) wait_dma<L_LATE>(left);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- COMPUTE phase
    __builtin_amdgcn_s_setprio(1);
#ifdef SP_GEMM_EXPERIMENTS
    if constexpr (DMA_IN_COMPUTE) {
      char *dsa = smem + stage_slot * STAGE + wave * 1024;
      char *dsb = dsa + A_BYTES;
      // pieces this wave owns: A pieces i < A_LOADS_HI or wave < A_SPLIT; B pieces j < B_LOADS_HI or wave < B_SPLIT
      int pieces = 0;
      if (issue) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) if (i < A_LOADS_HI || wave < A_SPLIT) pieces |= 1 << i;
#pragma unroll
        for (int j = 0; j < B_LOADS_LO; ++j) if (j < B_LOADS_HI || wave < B_SPLIT) pieces |= 1 << (A_LOADS + j);
      }
      mfma_block<TN, TM, A_LOADS, B_LOADS_LO, NWV>(acc, fw, fa, aptr, astep, bptr, dsa, dsb,
                                              __builtin_amdgcn_readfirstlane(pieces));
      if (issue) {
        stage_slot = stage_slot + 1 == PSTAGES ? 0 : stage_slot + 1;
        ++staged; ++in_tap;
      }
    } else if constexpr ((dbg & 2) != 0) {
#pragma unroll
      for (int i = 0; i < TN; ++i) asm volatile("" ::"v"(fw[i]));
#pragma unroll
      for (int j = 0; j < TM; ++j) asm volatile("" ::"v"(fa[j]));
    } else
#endif
    {
#pragma unroll
      for (int i = 0; i < TN; ++i) {
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if (!late) wait_dma<L_EARLY>(min(PDIST - 1, max(nk - 2 - kt, 0)));
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (!late) __builtin_amdgcn_s_barrier();
  PP_TRACE_CLK(9);
  PP_TRACE(2);

  if constexpr (SPLITK) {                         // raw fp32 sums of this K slice; the epilogue runs in the reduce kernel
    float *slab = p.partial + (int64_t)kslice * p.m * p.n;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int64_t m = (int64_t)tile_m * PBM + wm * WTM + j * 16 + fr;
      if (m < p.m) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
          *(f32x4 *)(slab + m * p.n + tile_n * BN + wn * WTN + i * 16 + 4 * fq) = acc[i][j];
      }
    }
  } else if (p.geglu) {
    if constexpr (TN % 2 == 0) pp_epilogue<PBM, BN, TN, TM, WTN, WTM, true, NT>(p, acc, smem, tile_m, tile_n, wm, wn, tid, fr, fq);
  } else {
    pp_epilogue<PBM, BN, TN, TM, WTN, WTM, false, NT, (EXP & 2048) != 0>(p, acc, smem, tile_m, tile_n, wm, wn, tid, fr, fq);
  }
  PP_TRACE(3);
}

template <int BM, int BN, int EXP = 0>
int launch_pp(GemmArgs &a, hipStream_t s) {
  constexpr int NSTG = pp_stages(BM, BN);
  constexpr size_t ring = (size_t)NSTG * (BM + BN) * 64, tile = (size_t)BM * BN * 2;   // epilogue stages the fp16 tile
  constexpr size_t lds = ring > tile ? ring : tile;
  static_assert(lds <= 160 * 1024, "LDS per workgroup");
  static bool attr_set[SP_MAX_DEVICES] = {};
  if (int rc = sp_ensure_dyn_lds((const void *)gemm_pp_kernel<BM, BN, EXP>, (int)lds, attr_set, "sp_gemm_f16(pp)"))
    return rc;
  a.tiles_m = (a.m + BM - 1) / BM;
  a.tiles_n = a.n / BN;
  note_kernel((EXP & 128) ? "gemm_pp_kernel<%d, %d, %d> + splitk_reduce_kernel" : "gemm_pp_kernel<%d, %d, %d>", BM, BN, EXP);
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL((gemm_pp_kernel<BM, BN, EXP>), dim3(a.tiles_m * a.tiles_n * ((EXP & 128) ? a.ksplit : 1)),
                     dim3(BM == 128 ? 256 : 512), lds, s, a);
  SP_CHECK_LAUNCH("sp_gemm_f16(pp)");
  return SP_OK;
}

}  // namespace

int launch_pp(GemmArgs &a, int bm, int bn, hipStream_t s) {
  if (a.ksplit >= 1 && a.partial) return launch_pp<256, 256, 128>(a, s);   // raw fp32 sums per K slice (1 slice: sp_gemm_f32out_f16)
  // GroupNorm column sums of an output that residuals are added to: the copy-out of this instantiation also rebuilds the
  // final tile in LDS and sums its columns there (the no-residual case sums the accumulators in the default kernels)
  if (a.gn_part && (a.res1 || a.res2)) return bn == 256 ? launch_pp<256, 256, 2048>(a, s) : launch_pp<256, 320, 2048>(a, s);
  if (a.a2) return bn == 256 ? launch_pp<256, 256, 4096>(a, s) : launch_pp<256, 320, 4096>(a, s);   // (host: never with the above)
#ifdef SP_GEMM_EXPERIMENTS
  if (bm == 256 && bn == 256) switch (a.dbg) {
    case 1: return launch_pp<256, 256, 1>(a, s);
    case 2: return launch_pp<256, 256, 2>(a, s);
    case 3: return launch_pp<256, 256, 3>(a, s);
    case 65: return launch_pp<256, 256, 65>(a, s);
    case 67: return launch_pp<256, 256, 67>(a, s);
    case 4: return launch_pp<256, 256, 4>(a, s);
    default: break;
  }
  if (bm == 256 && bn == 320 && a.dbg == 4) return launch_pp<256, 320, 4>(a, s);
  if (bm == 192 && bn == 256 && a.dbg == 4) return launch_pp<192, 256, 4>(a, s);
  if (a.dbg == 16) {
    if (bm == 256 && bn == 256) return launch_pp<256, 256, 16>(a, s);
    if (bm == 256 && bn == 320) return launch_pp<256, 320, 16>(a, s);
    if (bm == 192 && bn == 256) return launch_pp<192, 256, 16>(a, s);
  }
  if (a.dbg == 32 && bm == 256 && bn == 320) return launch_pp<256, 320, 32>(a, s);
  if (a.dbg == 8) {
    if (bm == 256 && bn == 256) return launch_pp<256, 256, 8>(a, s);
    if (bm == 256 && bn == 320) return launch_pp<256, 320, 8>(a, s);
    if (bm == 192 && bn == 256) return launch_pp<192, 256, 8>(a, s);
  }
#endif
  if (bn == 256) {
    if (bm == 128) return launch_pp<128, 256>(a, s);
    if (bm == 192) return launch_pp<192, 256>(a, s);
    return launch_pp<256, 256>(a, s);
  }
  if (bm == 192) return launch_pp<192, 320>(a, s);
  return launch_pp<256, 320>(a, s);
}

}  // namespace spgemm

#ifdef SP_GEMM_EXPERIMENTS
extern "C" int sp_debug_pp_trace(long long *host, int n_wgs) {
  if (n_wgs > spgemm::PP_TRACE_WGS) n_wgs = spgemm::PP_TRACE_WGS;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(spgemm::g_pp_trace),
                                  sizeof(long long) * n_wgs * spgemm::PP_TRACE_SLOTS);
}
#endif

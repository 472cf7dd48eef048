// Persistent-stream implicit-GEMM for gfx950: the K loop of gemm_pp.hip (256/192-row tiles, 8 waves, 4-deep LDS ring
// of 32-channel K-steps fed by LDS-DMA, ping-pong READ/COMPUTE phases) wrapped in a tile loop, for contractions with
// many tiles per CU and a short K (K <= 1280 is 40 % of the GEMM time of an SVD UNet step).  What changes:
//
//  * One workgroup per CU walks tiles b, b+G, b+2G, ...  The LDS-DMA stream never stops at a tile boundary: during
//    the last three K-steps of a tile the waves already issue the first three K-steps of their NEXT tile, so a tile
//    starts with its operands in LDS instead of paying launch gap + index setup + first-operand latency (~6 us of a
//    15-25 us tile in the one-tile-per-workgroup kernel).
//  * The epilogue does not go through LDS (the ring is busy with the next tile) and has no barrier: each lane
//    exchanges one register pair with its neighbour 16 lanes away (v_permlane16_swap) so that it owns 8 consecutive
//    output channels of one row, adds bias / residuals in fp32 (one rounding) and writes 16 bytes; a wave
//    instruction covers 16 rows x 64 contiguous bytes, which the memory system absorbs at the rate of whole-row
//    stores (tools/store_probe.hip: 5.6-6.1 vs 6.0-6.4 TB/s chip-wide).
//  * Bias vectors (and a folded LayerNorm's row sums / row statistics) arrive by LDS-DMA as well (three 1-KiB pieces
//    per wave per tile, read back with ds_read), so no
//    ordinary load sits in front of the stream: hipcc drains vmcnt to 0 before the first use of a VGPR load while
//    LDS-DMA is in flight.
//  * vmcnt bookkeeping.  LDS-DMA, loads and stores retire in issue order, so "K-step g+1 has landed" =
//    vmcnt(younger operations).  Behind a tile boundary the younger operations include the previous tile's stores and
//    this tile's two bias pieces; stores are raw buffer stores (rows past M are dropped by the range check), never
//    branched around, so their count per wave is a compile-time constant.
//
// Same GemmArgs contract as gemm_pp.hip / gemm.hip; bias2 must select ONE row per tile (the host checks).
#include "gemm_args.h"

namespace spgemm {
namespace {

constexpr int SBK = 32, SSTAGES = 4, SDIST = 3;
typedef unsigned v4u __attribute__((__vector_size__(4 * sizeof(unsigned))));

#ifdef SP_GEMM_EXPERIMENTS
// per-workgroup time split (100 MHz clock), waves 0 and 4: [start, first-barrier, sum K loop, sum realign, sum epilogue, end, tiles, -]
constexpr int PS_TRACE_WGS = 256, PS_TRACE_SLOTS = 16;
__device__ long long g_ps_trace[PS_TRACE_WGS * PS_TRACE_SLOTS];
#define PS_NOW() ((long long)wall_clock64())
#define PS_TRACE_DECL long long ps_t = 0, ps_loop = 0, ps_align = 0, ps_epi = 0, ps_start = PS_NOW(), ps_first = 0
#define PS_MARK(acc) do { const long long n__ = PS_NOW(); acc += n__ - ps_t; ps_t = n__; } while (0)
// per-K-step stamps of this workgroup's SECOND tile (through LDS: a global store would count in vmcnt)
constexpr int PS_STEP_SLOTS = 64;
__device__ long long g_ps_steps[PS_TRACE_WGS * 2 * PS_STEP_SLOTS];
#define PS_STEP(idx)                                                                                     \
  do {                                                                                                   \
    if (ps_tile == 1 && (tid == 0 || tid == 256) && (idx) < PS_STEP_SLOTS)                               \
      ((long long *)(smem + RING + 8 * 3072))[(tid ? PS_STEP_SLOTS : 0) + (idx)] = PS_NOW();             \
  } while (0)
#else
#define PS_STEP(idx) do {} while (0)
#define PS_NOW() 0
#define PS_TRACE_DECL
#define PS_MARK(acc) do {} while (0)
#endif

__device__ __forceinline__ int swz4(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// s_waitcnt vmcnt(n) for a wave-uniform n that is a compile-time constant at every (inlined) call site
__device__ __forceinline__ void wait_count(int n) {
#define W1(x) case x: wait_vm<x>(); break;
#define W8(x) W1(x) W1(x + 1) W1(x + 2) W1(x + 3) W1(x + 4) W1(x + 5) W1(x + 6) W1(x + 7)
  switch (n) { W8(0) W8(8) W8(16) W8(24) W8(32) W8(40) W8(48) W8(56) default: wait_vm<0>(); break; }
#undef W8
#undef W1
}
// at most `young` K-steps of this wave's DMA (L instructions each) plus `extra` other operations may stay outstanding;
// extra_kind: 0 none, 1 = X1 (first tile: bias pieces), 2 = X2 (later tiles: stores + bias pieces)
template <int L, int X1, int X2>
__device__ __forceinline__ void wait_stream(int young, int extra_kind) {
  wait_count(young * L + (extra_kind == 0 ? 0 : extra_kind == 1 ? X1 : X2));
}

__device__ __forceinline__ unsigned pack_h2(float a, float b) {
  const f16x2 h = {(f16)a, (f16)b};
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ f32x2 unpack_h2(unsigned u) {
  const f16x2 h = __builtin_bit_cast(f16x2, u);
  return (f32x2){(float)h[0], (float)h[1]};
}
// lanes of odd 16-lane rows give `a`, lanes of even rows give `b`; each receives the other's (v_permlane16_swap)
__device__ __forceinline__ void swap16(unsigned &a, unsigned &b) {
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}

// LDS-DMA piece with the address as (wave-uniform 64-bit base, 32-bit lane offset): the saddr form, written out because hipcc
// otherwise keeps the lane offsets zero-extended in register pairs and adds the base per piece (whole-line pairs issue from
// four bases: 8-16 registers the 256 x 256 tiles do not have).  M0 = LDS byte address of the piece.
__device__ __forceinline__ void glds16_s(const char *sbase, unsigned voff, char *lds_dst) {
  const unsigned l = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds_dst;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff), "s"(sbase), "s"(l) : "memory", "m0");
}

template <int BM, int BN, int WM, int WN, bool GEGLU, int VAR>
__global__ __launch_bounds__(512, 2) void gemm_ps_kernel(const GemmArgs p) {
  static_assert(WM * WN == 8, "eight waves");
  constexpr bool PAIR = (VAR & 1) != 0;                    // K-steps staged in pairs (see stage2)
  constexpr bool FULL = (VAR & 2) != 0;                    // ... as ONE 64-deep image of whole 128-byte lines (see stage2)
  static_assert(!FULL || PAIR, "whole-line pieces cover two K-steps");
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 16, TN = WTN / 16;
  constexpr int TNO = GEGLU ? TN / 2 : TN;                 // output sub-tiles (16 columns) per wave
  static_assert(TNO % 2 == 0, "output sub-tiles are written in pairs");
  constexpr int WTNO = GEGLU ? WTN / 2 : WTN, BNO = GEGLU ? BN / 2 : BN;
  constexpr int NSTORE = TM * (TNO / 2);                   // buffer stores per wave per tile
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES, RING = SSTAGES * STAGE;
  constexpr int A_PIECES = BM / 16, B_PIECES = BN / 16;   // 1-KiB DMA pieces (16 rows x 64 B), piece = j*8 + wave
  constexpr int A_LOADS = (A_PIECES + 7) / 8, A_LOADS_HI = A_PIECES / 8, A_SPLIT = A_PIECES % 8 ? A_PIECES % 8 : 8;
  constexpr int B_LOADS = (B_PIECES + 7) / 8, B_LOADS_HI = B_PIECES / 8, B_SPLIT = B_PIECES % 8 ? B_PIECES % 8 : 8;
  static_assert((A_SPLIT == 8 || A_SPLIT == 4) && (B_SPLIT == 8 || B_SPLIT == 4), "wave halves must have uniform DMA counts");
  constexpr int L_EARLY = A_LOADS + B_LOADS;
  constexpr int L_LATE = (A_SPLIT == 8 ? A_LOADS : A_LOADS_HI) + (B_SPLIT == 8 ? B_LOADS : B_LOADS_HI);
  constexpr int NBIAS = 3;                                 // 1-KiB side pieces per wave per tile: bias, bias2 row / LayerNorm
                                                           // row sums, LayerNorm (mean, rstd) of the wave's rows
  static_assert(2 * L_EARLY + NSTORE + NBIAS < 64, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef SP_GEMM_EXPERIMENTS
  // desynchronisation experiment: workgroups start in (stagger >> 8) phase groups, (stagger & 255) us apart
  if (p.stagger > 0) {
    const int groups = max(p.stagger >> 8, 1), d = (((int)blockIdx.x >> 3) % groups) * (p.stagger & 255);
    for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(30);   // 30 * 64 cycles ~ 1 us at 1.9 GHz
  }
#endif
  PS_TRACE_DECL;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const bool late = wave >= 4;
  const int fr = lane & 15, fq = lane >> 4;
  char *const bias_lds = smem + RING + wave * (NBIAS * 1024);

  // ---------------------------------------------------------------- tile walk
  constexpr int GM = 4;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int nmy = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // >= 1 (grid <= ntiles)
  auto decode = [&](int vt, int &tm, int &tn) {
    const int t = xcd_remap(vt, ntiles);
    const int per_group = GM * p.tiles_n;
    const int group = t / per_group;
    const int first_m = group * GM;
    const int gsz = min(p.tiles_m - first_m, GM);
    const int in_group = t - group * per_group;
    tn = in_group / gsz;
    tm = first_m + (in_group - tn * gsz);
  };

  // ---------------------------------------------------------------- producer (LDS-DMA stream) state
  // Source address of a DMA piece = wave-uniform 64-bit base (tile, K-step; advanced with scalar adds) + a per-lane
  // 32-bit byte offset that only depends on the lane's row inside the tile: moving the stream to the next tile is a
  // handful of scalar instructions.  Rows past M read row M-1 instead (their outputs are never stored).
  const int lrow = lane >> 2, lchunk = lane & 3;  // 16 rows x 4 chunks per 1-KiB piece
  const int nk = p.k >> 5;                        // K-steps per tile (host: nk >= 8)
  // FULL: a piece is 8 rows x 128 B (lane l: row l >> 3, LDS chunk l & 7 holding the line's chunk (l & 7) ^ (row & 7)); piece
  // i*8 + wave covers rows (i*8 + wave)*8.. of the tile's first half, its partner the same rows of the second half
  const int frow = lane >> 3, fchunk = ((lane & 7) ^ (lane >> 3)) * 8;
  unsigned a_off[A_LOADS], b_off[B_LOADS];
#pragma unroll
  for (int j = 0; j < B_LOADS; ++j) {
    if constexpr (FULL) {
      const int r = (j * 8 + wave) * 8 + frow;
      b_off[j] = (unsigned)(((r < BN / 2 ? r : 0) * p.k + fchunk) * 2);
    } else {
      const int r = (j * 8 + wave) * 16 + lrow;
      b_off[j] = (unsigned)(((r < BN ? r : 0) * p.k + (lchunk ^ swz4(r)) * 8) * 2);
    }
  }
  const char *a_base, *b_base;
  auto set_tile = [&](int tm, int tn) {
    const int rows_left = p.m - tm * BM;          // >= 1
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      if constexpr (FULL) {
        const int r = (i * 8 + wave) * 8 + frow;
        a_off[i] = (unsigned)(((r < BM / 2 ? r : 0) * (int)p.lda + fchunk) * 2);   // (host: m % BM == 0, no row to clamp)
      } else {
        const int r = (i * 8 + wave) * 16 + lrow;
        a_off[i] = (unsigned)((min(r < BM ? r : 0, rows_left - 1) * (int)p.lda + (lchunk ^ swz4(r)) * 8) * 2);
      }
    }
    a_base = (const char *)p.a + (int64_t)tm * BM * p.lda * 2;
    b_base = (const char *)p.w + (int64_t)tn * BN * p.k * 2;
  };
  int p_slot = 0;
  auto stage = [&]() {                            // LDS-DMA of the stream's next K-step
    char *sa = smem + p_slot * STAGE;
    char *sb = sa + A_BYTES;
    p_slot = p_slot + 1 == SSTAGES ? 0 : p_slot + 1;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i)
      if (i < A_LOADS_HI || wave < A_SPLIT)       // wave-uniform
        glds16(a_base + a_off[i], sa + (i * 8 + wave) * 1024);
#pragma unroll
    for (int j = 0; j < B_LOADS; ++j)
      if (j < B_LOADS_HI || wave < B_SPLIT)       // wave-uniform
        glds16(b_base + b_off[j], sb + (j * 8 + wave) * 1024);
    a_base += SBK * 2;
    b_base += SBK * 2;
  };

  // PAIR: two K-steps at a time, piece by piece -- a K-step reads 64 of the 128 bytes of every line it touches, and
  // with the other half requested one K-step (32 KiB of other lines through a 32 KiB L1) later every line travels from
  // L2 twice.  Back to back the second half hits in L1: tools/dma_probe.hip, all CUs on L2-resident rows, 63 -> 104 GB/s
  // per CU.  The kernel is not bound there (+0-4 %), see DESIGN.md.  (nk is even: the host checks.)
  // FULL (round 5): the pair of slots holds ONE image of 64-deep rows, [A: BM x 128 B | B: BN x 128 B], and a piece is 8 whole
  // lines instead of 16 half lines -- half the line requests per KiB on the texture path (tools/experiments/gemm_w2_probe.hip:
  // -5..7 % on a K loop against the same loop fed with half-line pieces).  Same number of pieces per wave and pair.
  auto stage2 = [&]() {
    char *sa = smem + p_slot * STAGE;              // p_slot is 0 or 2
    p_slot ^= 2;
    if constexpr (FULL) {
      char *sb = sa + 2 * A_BYTES;
      const char *b_hi = b_base + (int64_t)(BN / 2) * p.k * 2, *a_hi = a_base + (int64_t)(BM / 2) * p.lda * 2;
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i)
        if (i < A_LOADS_HI || wave < A_SPLIT) {
          glds16_s(a_base, a_off[i], sa + (i * 8 + wave) * 1024);
          glds16_s(a_hi, a_off[i], sa + A_BYTES + (i * 8 + wave) * 1024);
        }
#pragma unroll
      for (int j = 0; j < B_LOADS; ++j)
        if (j < B_LOADS_HI || wave < B_SPLIT) {
          glds16_s(b_base, b_off[j], sb + (j * 8 + wave) * 1024);
          glds16_s(b_hi, b_off[j], sb + B_BYTES + (j * 8 + wave) * 1024);
        }
    } else {
      char *sb = sa + A_BYTES;
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i)
        if (i < A_LOADS_HI || wave < A_SPLIT) {
          glds16(a_base + a_off[i], sa + (i * 8 + wave) * 1024);
          glds16(a_base + a_off[i] + SBK * 2, sa + STAGE + (i * 8 + wave) * 1024);
        }
#pragma unroll
      for (int j = 0; j < B_LOADS; ++j)
        if (j < B_LOADS_HI || wave < B_SPLIT) {
          glds16(b_base + b_off[j], sb + (j * 8 + wave) * 1024);
          glds16(b_base + b_off[j] + SBK * 2, sb + STAGE + (j * 8 + wave) * 1024);
        }
    }
    a_base += SBK * 4;
    b_base += SBK * 4;
  };

  int c_tm, c_tn;                                  // consumer's tile
  decode((int)blockIdx.x, c_tm, c_tn);
  set_tile(c_tm, c_tn);
  if constexpr (PAIR) {
    stage2(); stage2();
  } else {
#pragma unroll
    for (int s = 0; s < SSTAGES; ++s) stage();     // the whole ring: K-steps 0..3 (nk >= 8)
  }

  // fragment rows are (multiple of 16) + fr.  FULL: K-step ks of the pair reads chunk (4 * ks + fq) ^ (row & 7) of its
  // 128-byte row: the offsets below are those of the even K-step, the odd one's are the same ^ 64
  const int rd_chunk = FULL ? (fq ^ (fr & 7)) << 4 : (fq ^ swz4(fr)) << 4;
  constexpr int ROWB = FULL ? 128 : 64;
  // one address per operand and lane; the sub-tiles are 16 rows (immediate offsets) apart, and ^ 64 commutes with them
  const int offw0 = (FULL ? 2 * A_BYTES : A_BYTES) + (wn * WTN + fr) * ROWB + rd_chunk;
  const int offa0 = (wm * WTM + fr) * ROWB + rd_chunk;

  // output addressing of this lane: even 16-lane rows own 8 channels of the first sub-tile of a pair, odd rows of the second
  const int ocol = (fq & 1) * 16 + (fq >> 1) * 8;  // column of the 16-byte piece inside a pair of sub-tiles (32 columns)
#ifdef SP_GEMM_EXPERIMENTS
  // timing-only probe (dbg & 32): zero records -> the range check drops every store, the instruction stream stays
  const int d_bytes = (p.dbg & 32) ? 0 : (int)min((int64_t)p.m * p.ldd * 2, (int64_t)0x7fffffff);
#else
  const int d_bytes = (int)min((int64_t)p.m * p.ldd * 2, (int64_t)0x7fffffff);
#endif
  const __amdgpu_buffer_rsrc_t d_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.d, 0, d_bytes, 0x00020000);

  if constexpr (PAIR) {
    if (late) wait_vm<2 * L_LATE>(); else wait_vm<2 * L_EARLY>();  // K-steps 0 and 1 landed (this wave's part)
  } else {
    if (late) wait_vm<3 * L_LATE>(); else wait_vm<3 * L_EARLY>();  // K-step 0 landed (this wave's part)
  }
  __builtin_amdgcn_s_barrier();
#ifdef SP_GEMM_EXPERIMENTS
  ps_first = ps_t = PS_NOW();
#endif

  int read_slot = 0;
  f32x4 acc[TN][TM];
  // One K-step: READ phase (operand fragments of K-step kt to registers, LDS-DMA of the stream's next K-step, wait
  // until the stream's K-step kt+1 has landed) | barrier | COMPUTE phase | barrier.  `young`: K-steps issued beyond
  // kt+1; `xk`: what else is younger than K-step kt+1 (0 nothing, 1 bias pieces, 2 previous tile's stores + bias pieces).
#ifdef SP_GEMM_EXPERIMENTS
  int ps_tile = 0, ps_kidx = 0;
#endif
  auto kstep = [&](bool issue, int young, int xk, bool first) {
#ifdef SP_GEMM_EXPERIMENTS
    PS_STEP(ps_kidx); ++ps_kidx;
#endif
    const char *sa = smem + (FULL ? read_slot & ~1 : read_slot) * STAGE;
    const int odd = FULL ? (read_slot & 1) << 6 : 0;
    read_slot = read_slot + 1 == SSTAGES ? 0 : read_slot + 1;
    f16x8 fw[TN], fa[TM];
#pragma unroll
    for (int i = 0; i < TN; ++i) fw[i] = *(const f16x8 *)(sa + (FULL ? offw0 ^ odd : offw0) + i * 16 * ROWB);
#pragma unroll
    for (int j = 0; j < TM; ++j) fa[j] = *(const f16x8 *)(sa + (FULL ? offa0 ^ odd : offa0) + j * 16 * ROWB);
    __builtin_amdgcn_sched_barrier(0);
    if (first) {
      // this tile's bias vectors (its epilogue is >= 8 K-steps away; the previous tile's epilogue has read its own)
      const int c0 = min(c_tn * BN + wn * WTN + lane * 4, p.n - 4);
      const float *b1 = p.bias ? p.bias + c0 : (const float *)p.zero;
      const float *b2 = (const float *)p.zero;
      if (p.bias2) b2 = p.bias2 + ((int64_t)(c_tm * BM) / p.bias2_rows) * p.ldb2 + c0;
      if (p.ln_stats) b2 = p.ln_colsum + c0;        // folded LayerNorm: the second piece carries the weight's row sums
      const char *b3 = p.zero;
      if (p.ln_stats)   // (mean, rstd) of this wave's WTM rows: 8 bytes per row, two rows per lane; clamped at the last row pair
        b3 = (const char *)p.ln_stats + min(((int64_t)c_tm * BM + wm * WTM) * 8 + lane * 16, (int64_t)p.m * 8 - 16);
      glds16(b1, bias_lds);
      glds16(b2, bias_lds + 1024);
      glds16(b3, bias_lds + 2048);
    }
    if (issue) { if constexpr (PAIR) stage2(); else stage(); }
    if (late && young >= 0) wait_stream<L_LATE, NBIAS, NSTORE + NBIAS>(young, xk);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
      for (int j = 0; j < TM; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (!late && young >= 0) wait_stream<L_EARLY, NBIAS, NSTORE + NBIAS>(young, xk);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  for (int ti = 0; ti < nmy; ++ti) {
    // Waves 4-7 run one phase behind waves 0-3 inside a tile (w and w+4 share a SIMD: one computes while the other
    // fetches operands); the halves are re-aligned around the epilogue so that both run it at the same time.
    if (late) __builtin_amdgcn_s_barrier();
    PS_MARK(ps_align);
#ifdef SP_GEMM_EXPERIMENTS
    ps_tile = ti; ps_kidx = 0;
#endif
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int xk0 = ti == 0 ? 1 : 2;
    const bool more = ti + 1 < nmy;

    // K-steps 0 .. nk-4: the stream stays inside this tile.  K-steps 0..3 are already in the ring (K-step 3 was issued
    // ahead of the previous tile's stores), so K-step kt issues kt+3 from kt = 1 on; while K-step 3 is the youngest
    // load the previous tile's stores and this tile's bias pieces sit behind it in the queue (xk0).
    int n_tm = c_tm, n_tn = c_tn;
    if constexpr (PAIR) {
      // Pairs (0,1) and (2,3) are in the ring (issued ahead of the previous tile's stores).  Even K-step k issues the
      // pair (k+2, k+3) into the slots of k-2 and k-1; odd K-step k waits for the whole pair (k+1, k+2) -- nothing
      // younger is in the queue then, except behind the tile boundary: the previous tile's stores and the bias pieces.
      kstep(false, -1, 0, true);
      kstep(false, 0, xk0, false);
      for (int kt = 2; kt < nk - 2; kt += 2) {
        kstep(true, -1, 0, false);
        kstep(false, 0, 0, false);
      }
      if (more) {
        decode((int)blockIdx.x + (ti + 1) * (int)gridDim.x, n_tm, n_tn);
        set_tile(n_tm, n_tn);
      }
      kstep(more, -1, 0, false);                   // the next tile's pair (0,1)
      kstep(false, 0, 0, false);
    } else {
      kstep(false, 2, xk0, true);
      kstep(true, 2, xk0, false);
      kstep(true, 2, xk0, false);
      for (int kt = 3; kt < nk - SDIST; ++kt) kstep(true, 2, 0, false);
      // K-steps nk-3 .. nk-1: the stream moves on to the first three K-steps of this workgroup's next tile
      if (more) {
        decode((int)blockIdx.x + (ti + 1) * (int)gridDim.x, n_tm, n_tn);
        set_tile(n_tm, n_tn);
      }
      kstep(more, more ? 2 : 1, 0, false);
      kstep(more, more ? 2 : 0, 0, false);
      kstep(more, more ? 2 : 0, 0, false);
    }
    PS_MARK(ps_loop);
    PS_STEP(ps_kidx);
    if (!late) __builtin_amdgcn_s_barrier();
    PS_MARK(ps_align);
    PS_STEP(ps_kidx + 1);
    // The ring slots of this tile's last K-step(s) are free now (every wave has read them): the next tile's K-step 3
    // (PAIR: K-steps 2 and 3) goes out BEFORE the stores.  Loads retire in order behind older stores, so a load issued
    // after them would only count as landed once the stores have been acknowledged.
    if (more) { if constexpr (PAIR) stage2(); else stage(); }

    // ---------------------------------------------------------------- epilogue (no LDS ring use, no barrier)
    // acc[i][j][r]: channel n = wn*WTN + i*16 + 4*fq + r, row m = wm*WTM + j*16 + fr of tile (c_tm, c_tn)
    {
      const int64_t mrow0 = (int64_t)c_tm * BM + wm * WTM + fr;
      const int col0 = c_tn * BNO + wn * WTNO + ocol;        // + pair*32
      uint4 q[TM][TNO / 2];                                  // residual pieces, then the packed output, in the FINAL
                                                             // layout (8 consecutive channels per lane)
      // ---- folded LayerNorm: acc <- rstd[m]*(acc - mean[m]*colsum[n]); its bias is added below like any other
      const bool ln = p.ln_stats != nullptr;
      if (ln) {
        f32x2 st[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) st[j] = *(const f32x2 *)(bias_lds + 2048 + (j * 16 + fr) * 8);
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const f32x4 cs = *(const f32x4 *)(bias_lds + 1024 + (i * 16 + 4 * fq) * 4);
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[i][j] = (acc[i][j] - st[j][0] * cs) * st[j][1];
        }
      }
      // ---- (acc + bias) * scale [GEGLU: value * gelu(gate)] -> va/vb kept in the accumulator registers
      const float osc = p.oscale;
#pragma unroll
      for (int o = 0; o < TNO / 2; ++o) {
        f32x4 bs[GEGLU ? 4 : 2];
#pragma unroll
        for (int t = 0; t < (GEGLU ? 4 : 2); ++t) {
          const int i = (GEGLU ? 4 : 2) * o + t;
          bs[t] = *(const f32x4 *)(bias_lds + (i * 16 + 4 * fq) * 4);
          if (!ln) bs[t] += *(const f32x4 *)(bias_lds + 1024 + (i * 16 + 4 * fq) * 4);
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          if constexpr (GEGLU) {
            const f32x4 h0 = (acc[4 * o][j] + bs[0]) * osc, g0 = acc[4 * o + 1][j] + bs[1];
            const f32x4 h1 = (acc[4 * o + 2][j] + bs[2]) * osc, g1 = acc[4 * o + 3][j] + bs[3];
            const f32x2 ga = gelu2_f((f32x2){g0[0], g0[1]}), gb = gelu2_f((f32x2){g0[2], g0[3]});
            const f32x2 gc = gelu2_f((f32x2){g1[0], g1[1]}), gd = gelu2_f((f32x2){g1[2], g1[3]});
            acc[4 * o][j] = (f32x4){h0[0] * ga[0], h0[1] * ga[1], h0[2] * gb[0], h0[3] * gb[1]};
            acc[4 * o + 1][j] = (f32x4){h1[0] * gc[0], h1[1] * gc[1], h1[2] * gd[0], h1[3] * gd[1]};
          } else {
            acc[2 * o][j] = (acc[2 * o][j] + bs[0]) * osc;
            acc[2 * o + 1][j] = (acc[2 * o + 1][j] + bs[1]) * osc;
          }
        }
      }
      // ---- residuals: all loads of one tensor are issued together, folded into the accumulators in fp32
      auto add_residual = [&](const f16 *res, int64_t ldr, float s) {
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)res, 0, (int)min((int64_t)p.m * ldr * 2, (int64_t)0x7fffffff), 0x00020000);
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int o = 0; o < TNO / 2; ++o)
            q[j][o] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                r_rsrc, (int)(((mrow0 + j * 16) * ldr + col0 + o * 32) * 2), 0, 0));
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int o = 0; o < TNO / 2; ++o) {
            uint4 t = q[j][o];
            swap16(t.x, t.z); swap16(t.y, t.w);           // back to the accumulator layout: (x,y) first sub-tile, (z,w) second
            const f32x2 a0 = unpack_h2(t.x), a1 = unpack_h2(t.y), b0 = unpack_h2(t.z), b1 = unpack_h2(t.w);
            constexpr int S = GEGLU ? 4 : 2;
            f32x4 &va = acc[S * o][j], &vb = acc[S * o + 1][j];
            va[0] += s * a0[0]; va[1] += s * a0[1]; va[2] += s * a1[0]; va[3] += s * a1[1];
            vb[0] += s * b0[0]; vb[1] += s * b0[1]; vb[2] += s * b1[0]; vb[3] += s * b1[1];
          }
      };
      if (p.res1) add_residual(p.res1, p.ldr1, p.r1scale);
      if (p.res2) add_residual(p.res2, p.ldr2, p.r2scale);
      // ---- pack, exchange with the neighbour 16 lanes away, store (rows past M fall outside the buffer: dropped)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int o = 0; o < TNO / 2; ++o) {
          constexpr int S = GEGLU ? 4 : 2;
          const f32x4 va = acc[S * o][j], vb = acc[S * o + 1][j];
          uint4 out;
          out.x = pack_h2(va[0], va[1]); out.y = pack_h2(va[2], va[3]);
          out.z = pack_h2(vb[0], vb[1]); out.w = pack_h2(vb[2], vb[3]);
          swap16(out.x, out.z); swap16(out.y, out.w);     // even rows: 8 channels of the first sub-tile, odd rows: of the second
          q[j][o] = out;
        }
      __builtin_amdgcn_sched_barrier(0);
      // Non-temporal stores (aux = 2).  Loads and stores retire through ONE in-order queue per wave, so the next tile's
      // K-step 4 counts as landed only once this tile's stores have been acknowledged; streaming stores are acknowledged
      // sooner and leave the XCD's L2 to the operands.  Same-process A/B (tools/bench_dbg.py, bit 256 = the OLD default
      // stores): GEGLU FF1 258,048 x 2,560 x 320 534.7 -> 516.3 us, 64,512 x 5,120 x 640 419.8 -> 408.7, plain
      // 64,512 x 2,560 x 640 249.4 -> 215.3, 16,128 x 10,240 x 1,280 352.1 -> 350.9 (profiles/r04_nt_stores_ab.txt).
#ifdef SP_GEMM_EXPERIMENTS
      if (p.dbg & 256) {                                  // default-policy stores: the A/B's other arm
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int o = 0; o < TNO / 2; ++o)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, q[j][o]), d_rsrc,
                                                   (int)(((mrow0 + j * 16) * p.ldd + col0 + o * 32) * 2), 0, 0);
      } else
#endif
      {
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int o = 0; o < TNO / 2; ++o)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, q[j][o]), d_rsrc,
                                                 (int)(((mrow0 + j * 16) * p.ldd + col0 + o * 32) * 2), 0, 2);
      }
    }
    c_tm = n_tm; c_tn = n_tn;
    PS_MARK(ps_epi);
    PS_STEP(ps_kidx + 2);
  }
#ifdef SP_GEMM_EXPERIMENTS
  if ((tid == 0 || tid == 256) && blockIdx.x < PS_TRACE_WGS) {
    for (int i = 0; i < PS_STEP_SLOTS; ++i)
      g_ps_steps[(blockIdx.x * 2 + (tid ? 1 : 0)) * PS_STEP_SLOTS + i] = ((long long *)(smem + RING + 8 * 3072))[(tid ? PS_STEP_SLOTS : 0) + i];
    long long *t = g_ps_trace + blockIdx.x * PS_TRACE_SLOTS + (tid ? 8 : 0);
    t[0] = ps_start; t[1] = ps_first; t[2] = ps_loop; t[3] = ps_align; t[4] = ps_epi; t[5] = PS_NOW(); t[6] = nmy;
  }
#endif
}

template <int BM, int BN, int WM, int WN, bool GEGLU, int VAR>
int launch_ps_t(GemmArgs &a, hipStream_t s) {
#ifdef SP_GEMM_EXPERIMENTS
  constexpr size_t lds = (size_t)SSTAGES * (BM + BN) * 64 + 8 * 3072 + 1024;   // + per-K-step stamps
#else
  constexpr size_t lds = (size_t)SSTAGES * (BM + BN) * 64 + 8 * 3072;
#endif
  static_assert(lds <= 160 * 1024, "LDS per workgroup");
  static bool attr_set[SP_MAX_DEVICES] = {};
  if (int rc = sp_ensure_dyn_lds((const void *)gemm_ps_kernel<BM, BN, WM, WN, GEGLU, VAR>, (int)lds, attr_set, "sp_gemm_f16(ps)"))
    return rc;
  a.tiles_m = (a.m + BM - 1) / BM;
  a.tiles_n = a.n / BN;
  const int ntiles = a.tiles_m * a.tiles_n;
  const int grid = ntiles < 256 ? ntiles : 256;       // one workgroup per CU
  note_kernel("gemm_ps_kernel<%d, %d, %d, %d, %s, %d>", BM, BN, WM, WN, GEGLU ? "true" : "false", VAR);
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL((gemm_ps_kernel<BM, BN, WM, WN, GEGLU, VAR>), dim3(grid), dim3(512), lds, s, a);
  SP_CHECK_LAUNCH("sp_gemm_f16(ps)");
  return SP_OK;
}

}  // namespace

bool ps_supported(const GemmArgs &a, int bm, int bn) {
  // 256-column tiles of 256 / 192 rows (2 x 4 waves), or 320-column tiles of 128 rows (4 x 2 waves, ten column
  // sub-tiles of two row sub-tiles per wave: 80 accumulator registers).  A 256 x 320 tile in that arrangement holds 160
  // accumulator registers and its residual prefetch does not fit the register file without spills (measured 0.6x the
  // ping-pong kernel in round 2).
  if (!((bn == 256 && (bm == 256 || bm == 192)) || (bn == 320 && bm == 128) || (bn == 192 && bm == 256))) return false;
  if ((bn == 320 || bn == 192) && a.geglu) return false;         // (value, gate) pairs of an odd number of sub-tile pairs
  if (a.n % bn) return false;
  if (a.mode != SP_A_LINEAR) return false;        // the tile loop streams plain rows (nn.Linear / 1x1 convolution)
  if (a.k < 8 * SBK) return false;
  if (a.n_store > 0) return false;
  if (a.ldd % 8 || (a.res1 && a.ldr1 % 8) || (a.res2 && a.ldr2 % 8)) return false;
  if ((int64_t)a.m * a.ldd * 2 >= 0x7fffffff) return false;
  // residuals are read through 32-bit buffer offsets as well (a column slice of a wider tensor has ldr > ldd)
  if (a.res1 && (int64_t)a.m * a.ldr1 * 2 >= 0x7fffffff) return false;
  if (a.res2 && (int64_t)a.m * a.ldr2 * 2 >= 0x7fffffff) return false;
  // folded LayerNorm: the rows' (mean, rstd) are fetched two rows per lane with the pair clamped to m*8-16; with an
  // odd m the pair (m-1, m) would be shifted to (m-2, m-1) and row m-1 normalised with its neighbour's statistics.
  // The ping-pong kernel loads them per row: let it take odd row counts.
  if (a.ln_stats && (a.m & 1)) return false;
  if (a.bias2 && a.bias2_rows < a.m && a.bias2_rows % bm) return false;   // a tile must not straddle two bias2 rows
  return true;
}

template <int VAR>
int launch_ps_v(GemmArgs &a, int bm, int bn, hipStream_t s) {
  if (bn == 320) return launch_ps_t<128, 320, 4, 2, false, VAR>(a, s);
  // 256 x 192 (4 x 2 waves of 64 x 96: 96 accumulator registers): widths that are multiples of 192 but not of 256 -- the
  // fused Q/K/V projections of the two outer levels (960 = 5 x 192, 1,920 = 10 x 192)
  if (bn == 192) return launch_ps_t<256, 192, 4, 2, false, VAR>(a, s);
  if (a.geglu) return bm == 192 ? launch_ps_t<192, 256, 2, 4, true, VAR>(a, s) : launch_ps_t<256, 256, 2, 4, true, VAR>(a, s);
  return bm == 192 ? launch_ps_t<192, 256, 2, 4, false, VAR>(a, s) : launch_ps_t<256, 256, 2, 4, false, VAR>(a, s);
}

int launch_ps(GemmArgs &a, int bm, int bn, hipStream_t s) {
  // K >= 640: K-steps staged in pairs (+3-4 % on the FF contractions of the 32,256- and 8,064-row levels; at K = 320 a
  // tile is 10 K-steps and the shorter prefetch distance of the pairs costs 4 %)
  bool pair = a.k >= 20 * SBK && (a.k / SBK) % 2 == 0;
  // pairs as one image of whole 128-byte lines (VAR 3): the activation rows must start on a line (lda % 64, base % 128)
  // and every tile be whole (the second half's rows are addressed from the first half's, without the clamp at row m-1)
  bool full = pair && a.m % bm == 0 && a.lda % 64 == 0 && a.k % 64 == 0 && ((uintptr_t)a.a & 127) == 0 && ((uintptr_t)a.w & 127) == 0;
#ifdef SP_GEMM_EXPERIMENTS
  if (a.dbg & 16) pair = false;
  if (a.dbg & 128) full = false;
#endif
  if (pair && full) return launch_ps_v<3>(a, bm, bn, s);
  return pair ? launch_ps_v<1>(a, bm, bn, s) : launch_ps_v<0>(a, bm, bn, s);
}

}  // namespace spgemm

#ifdef SP_GEMM_EXPERIMENTS
extern "C" int sp_debug_ps_steps(long long *host, int n_wgs) {
  if (n_wgs > spgemm::PS_TRACE_WGS) n_wgs = spgemm::PS_TRACE_WGS;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(spgemm::g_ps_steps), sizeof(long long) * n_wgs * 2 * spgemm::PS_STEP_SLOTS);
}
extern "C" int sp_debug_ps_trace(long long *host, int n_wgs) {
  if (n_wgs > spgemm::PS_TRACE_WGS) n_wgs = spgemm::PS_TRACE_WGS;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(spgemm::g_ps_trace), sizeof(long long) * n_wgs * spgemm::PS_TRACE_SLOTS);
}
#endif

// Attention kernels for gfx950 (head_dim 64, fp16 in / fp32 accumulate, no mask, no dropout).
//
// attn_spatial_kernel – flash-style forward over `seq` tokens of one frame.
//   Workgroup = 4 waves x 32 query rows = 128 query rows of one (frame, head); KV tiles of 64 keys
//   are brought into a 2-deep LDS ring by global_load_lds_dwordx4 with the 16-byte chunk XOR-swizzled
//   on the source side (K: (row>>1)&7 for conflict-free ds_read_b128 operand reads; V: bit 2 from
//   (row>>1)&1 for conflict-free ds_read_b64_tr_b16 transposed reads).
//   Scores are computed TRANSPOSED, S^T = K.Q^T with v_mfma_f32_32x32x16_f16, so each lane holds 32
//   keys of ONE query column: the softmax row reduction is 31 in-register max/adds plus one
//   cross-half exchange (wavefront-level reduction, no LDS), and the fp32 score accumulator converts
//   in place into the B operand of O^T += V^T.P^T (the accumulator-as-operand k-order
//   16s + 8(j>>2) + 4h + (j&3) is matched by the transposed V reads).
//   The kernel is VALU-issue bound (32 v_exp_f32 + ~120 other VALU per 16 MFMAs), so the per-score arithmetic is cut
//   to the exponential itself: Q is pre-multiplied by scale*log2(e) when its fragments are loaded (scores come out of
//   the MFMA in log2 units) and the score accumulator is INITIALISED to -m (the running row maximum), so that
//   p = exp2(acc) needs no subtract; only when a tile raises some row's maximum (rare after the first tiles) the
//   wave subtracts the increase and rescales O and l.
//
// attn_temporal_kernel – sequences run across frames (14 or 25 tokens) for each pixel/head.
//   One wave per (pixel, head); Q/K fragments are loaded straight from global memory in MFMA operand
//   layout (token (f,p) = row f*hw+p, so the (B*HW, F, C) permute of the reference never exists),
//   S^T via v_mfma_f32_16x16x32_f16, softmax across the 4 lane groups, P (still in accumulator
//   layout) is directly the A operand of v_mfma_f32_16x16x16_f16 for O = P.V.  HBM-bound.
#include "common.h"

namespace {

typedef short v4i16 __attribute__((__vector_size__(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f16x4 lds_tr16(const char *p) {
  v4i16 raw = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16 *)p);
  return __builtin_bit_cast(f16x4, raw);
}

constexpr float NEG_BIG = -3.0e38f;

// raw v_exp_f32 (2^x): arguments here are <= 0 and results below 2^-126 may flush to zero, so the
// range-reduction / denormal wrapper that exp2f() adds (6 extra VALU ops per element) is not needed.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_spatial_kernel(
    const f16 *__restrict__ q, const f16 *__restrict__ k, const f16 *__restrict__ v, f16 *__restrict__ o,
    int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int seq, int heads, float scale_log2e,
    const char *__restrict__ zero, const unsigned *__restrict__ only_flagged) {
  constexpr int KV = 64;
  constexpr int K_BYTES = KV * 128, STAGE = 2 * K_BYTES, RING = 2;
  __shared__ __attribute__((aligned(16))) char smem[RING * STAGE];
  // second pass of sp_attn_spatial_long_f16 (attention_long.hip): one word per 256 query rows of a (batch item, head),
  // non-zero = do these rows (uniform: a scalar load and branch)
  if (only_flagged && only_flagged[(int64_t)blockIdx.y * ((seq + 255) >> 8) + (blockIdx.x >> 1)] == 0) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.y;
  const int b = bh / heads, hd = bh - b * heads;
  const int r = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)b * seq;
  const int q0 = blockIdx.x * 128 + wave * 32;

  // Q fragments (B operand of S^T = K.Q^T): lane holds Q[q0 + r][16s + 8h .. +8], pre-multiplied by scale*log2(e)
  // (fp32 product, one rounding to fp16)
  f16x8 qf[4];
  {
    const bool ok = q0 + r < seq;
    const f16 *qp = ok ? q + (row0 + q0 + r) * ldq + hd * 64 + 8 * h : (const f16 *)zero;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f16x8 raw = *(const f16x8 *)(qp + (ok ? 16 * s : 0));
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[s][e] = (f16)((float)raw[e] * scale_log2e);
    }
  }

  // K/V staging: each wave moves 16 rows of K and of V per tile (2 x 2 LDS-DMA pieces of 8 rows).  Source address =
  // wave-uniform base of the tile (advanced with scalar adds) + a per-lane byte offset that is the same for every
  // tile; keys past `seq` (last tile only) re-read row seq-1: their scores are masked below and P = 0 meets a finite V.
  const int lrow = lane >> 3, lchunk = lane & 7;
  const char *kb = (const char *)(k + row0 * ldk + hd * 64);
  const char *vb = (const char *)(v + row0 * ldv + hd * 64);
  unsigned kofl[2], vofl[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * 16 + i * 8 + lrow;
    kofl[i] = (unsigned)((row * ldk + ((lchunk ^ ((row >> 1) & 7)) << 3)) * 2);
    vofl[i] = (unsigned)((row * ldv + ((lchunk ^ (((row >> 1) & 1) << 2)) << 3)) * 2);
  }
  const int64_t kstep = (int64_t)KV * ldk * 2, vstep = (int64_t)KV * ldv * 2;
  auto stage = [&](int tile, int buf) {
    char *sk = smem + buf * STAGE;
    char *sv = sk + K_BYTES;
    if ((tile + 1) * KV <= seq) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        glds16(kb + kofl[i], sk + (wave * 16 + i * 8) * 128);
        glds16(vb + vofl[i], sv + (wave * 16 + i * 8) * 128);
      }
    } else {                                        // ragged last tile
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wave * 16 + i * 8 + lrow;
        const int64_t back = max(tile * KV + row - (seq - 1), 0);     // rows to step back to stay inside the sequence
        glds16(kb + kofl[i] - back * ldk * 2, sk + (wave * 16 + i * 8) * 128);
        glds16(vb + vofl[i] - back * ldv * 2, sv + (wave * 16 + i * 8) * 128);
      }
    }
    kb += kstep;
    vb += vstep;
  };

  // operand read offsets
  int koff[2];   // K fragment row offsets (+ chunk term added per s)
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) koff[kt] = (kt * 32 + r) * 128;
  const int kswz = (r >> 1) & 7;
  // V transposed-read lane address pieces: group g = lane>>4, i = lane&15, q_ = i>>2, p = i&3
  const int g = lane >> 4, i16 = lane & 15, q_ = i16 >> 2, pp = i16 & 3;
  const int vrow_l = 4 * (g >> 1) + q_;                       // + key0 base (multiple of 8)
  const int vchunk_l = 2 * (g & 1) + (pp >> 1);               // + 4*dt
  const int vswz = (q_ >> 1) << 2;
  const int vbyte_l = (pp & 1) * 8;
  // lane part of the transposed-read address for d-tile 0 / 1; the key offset is a compile-time immediate
  const int vlane0 = vrow_l * 128 + ((vchunk_l ^ vswz) << 4) + vbyte_l;
  const int vlane1 = vrow_l * 128 + (((4 + vchunk_l) ^ vswz) << 4) + vbyte_l;

  f32x16 oacc[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) { oacc[0][e] = 0.f; oacc[1][e] = 0.f; }
  float m_run = 0.f;                         // running row maximum in log2 units (set by the first tile)
  // Row sums on the matrix pipe (the kernel is bound by vector ISSUE, the matrix pipe has slack): the packed P
  // registers of a k-step, read as the B operand of v_mfma_f32_16x16x32_f16, are column n = lane & 15 and k-block
  // lane >> 4; k-blocks 0 / 2 carry this wave's query column n (both key halves), k-blocks 1 / 3 query column n + 16.
  // With the constant selector A (rows 0 and 8: ones over k-blocks 0 and 2; rows 4 and 12: ones over k-blocks 1 and
  // 3; zero elsewhere) accumulator element 0 of EVERY lane is the sum over all 16 keys of the k-step for that lane's
  // own query column r -- 4 MFMAs of 16 cycles per tile instead of 32 fp32 adds, no cross-half exchange at the end,
  // and l sums exactly the fp16-rounded probabilities the numerator uses.
  f32x4 lacc = {0.f, 0.f, 0.f, 0.f};
  f16x8 lsel;
  {
    const int m16 = lane & 15, kb = lane >> 4;
    const bool one = ((m16 & 7) == 0 && (kb & 1) == 0) || ((m16 & 7) == 4 && (kb & 1) == 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) lsel[e] = one ? (f16)1.f : (f16)0.f;
  }
  f32x16 negm;                               // -m_run in all 16 accumulator positions: the C operand of the first score MFMA
#pragma unroll
  for (int e = 0; e < 16; ++e) negm[e] = 0.f;

  const int ntiles = (seq + KV - 1) / KV;
  // 2-deep K/V ring: tile t+1 is issued while tile t is consumed and must have landed at the end of tile t.  The
  // transposed V reads are inline asm: for the builtin hipcc puts an s_waitcnt vmcnt(0) in front of the first one
  // (LDS-DMA in flight), which waits for the NEXT tile in the middle of this one.
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int buf = 0;
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) stage(t + 1, buf ^ 1);
    const char *sk = smem + buf * STAGE;
    const char *sv = sk + K_BYTES;
    const char *sv0 = sv + vlane0, *sv1 = sv + vlane1;

    // ---- S^T = K.Q^T - m : two 32-key sub-tiles, accumulators start at minus the running maximum
    f32x16 sacc[2];
#pragma unroll
    for (int s = 0; s < 4; ++s) {      // k-step-major: both chains take -m as their C operand with a fresh destination
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const f16x8 kf = *(const f16x8 *)(sk + koff[kt] + (((2 * s + h) ^ kswz) << 4));
        sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], s == 0 ? negm : sacc[kt], 0, 0, 0);
      }
    }
    // mask keys beyond seq (only the last tile can be partial)
    if ((t + 1) * KV > seq) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = t * KV + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= seq) sacc[kt][e] = NEG_BIG;
        }
    }
    // ---- online softmax for query column r (keys split over the two lane halves)
    float mt = fmaxf(fmaxf(sacc[0][0], sacc[0][1]), sacc[0][2]);
#pragma unroll
    for (int e = 3; e < 31; e += 2) mt = fmaxf(fmaxf(mt, sacc[e >> 4][e & 15]), sacc[(e + 1) >> 4][(e + 1) & 15]);
    mt = fmaxf(mt, sacc[1][15]);
    {   // the other 32 keys of this query column live 32 lanes away: v_permlane32_swap (a VALU exchange; __shfl_xor is a
        // ds_bpermute round trip through the LDS crossbar with an lgkmcnt wait in the middle of the tile).  Inline asm:
        // with the builtin hipcc (ROCm 7.2) uses only the first result when both operands hold the same value.
      float ma = mt, mb = mt;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
      mt = fmaxf(ma, mb);                 // ma = {own lower, other lower}, mb = {other upper, own upper}
    }
    // mt = (tile maximum) - m_run.  The first tile defines the reference; later tiles raise it only when a score
    // exceeds it by more than RESCALE_LOG2 (p = 2^(s - m) <= 2^8 is exact in fp16 relative precision and l, O are fp32):
    // with 32 query columns per wave SOME column would otherwise move in most tiles and make the whole wave rescale.
    constexpr float RESCALE_LOG2 = 8.0f;
    const float delta = t == 0 ? mt : (mt > RESCALE_LOG2 ? mt : 0.f);
    if (__builtin_amdgcn_ballot_w64(delta != 0.f) != 0) {
      const float alpha = t == 0 ? 1.f : fast_exp2(-delta);   // (first tile: l and O are still zero; -delta may be huge)
      lacc[0] *= alpha;
#pragma unroll
      for (int e = 0; e < 16; ++e) { oacc[0][e] *= alpha; oacc[1][e] *= alpha; }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[kt][e] -= delta;
      m_run += delta;
#pragma unroll
      for (int e = 0; e < 16; ++e) negm[e] = -m_run;
    }
    // P -> fp16 pairwise (v_cvt_pk_f16_f32, round-to-nearest: one instruction per pair instead of two converts
    // and a pack)
    u32x4 pw[2][2];                                   // pw[kt][s] = the 8 fp16 P values of k-step s, as four packed pairs
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const f32x2 p = {fast_exp2(sacc[kt][e]), fast_exp2(sacc[kt][e + 1])};
        pw[kt][e >> 3][(e & 7) >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(p, f16x2));
      }

    // ---- O^T += V^T . P^T : the 8 transposed reads of one d-tile go out together (inline asm, own lgkmcnt wait)
    {
      const unsigned va[2] = {(unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)sv0,
                              (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)sv1};
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        u32x2 vr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)        // i = 4*kt + 2*s + e2: keys kt*32 + 16*s + 8*e2 ..
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vr[i]) : "v"(va[dt]), "n"(i * 8 * 128) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vr[0]), "+v"(vr[1]), "+v"(vr[2]), "+v"(vr[3]), "+v"(vr[4]), "+v"(vr[5]),
                     "+v"(vr[6]), "+v"(vr[7])::"memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const u32x4 vw = {vr[4 * kt + 2 * s][0], vr[4 * kt + 2 * s][1], vr[4 * kt + 2 * s + 1][0], vr[4 * kt + 2 * s + 1][1]};
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, vw), __builtin_bit_cast(f16x8, pw[kt][s]),
                                                             oacc[dt], 0, 0, 0);
            if (dt == 0)    // row sums of this k-step (see lsel)
              lacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(lsel, __builtin_bit_cast(f16x8, pw[kt][s]), lacc, 0, 0, 0);
          }
      }
    }
    // tile t+1 landed (this wave's pieces), every wave done reading tile t's slot before it is refilled next iteration
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    buf ^= 1;
  }

  // ---- finalize: O^T[d][q] / l ; lane holds d = 32dt + (e&3) + 8(e>>2) + 4h for query q0 + r
  const float inv = 1.0f / lacc[0];
  if (q0 + r < seq) {
    f16 *op = o + (row0 + q0 + r) * ldo + hd * 64 + 4 * h;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (f16)(oacc[dt][4 * c + e] * inv);
        *(f16x4 *)(op + 32 * dt + 8 * c) = w;
      }
  }
}

// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void attn_temporal_kernel(
    const f16 *__restrict__ q, const f16 *__restrict__ k, const f16 *__restrict__ v, f16 *__restrict__ o,
    int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int frames, int64_t hw, int heads,
    int64_t nproblems, float scale_log2e, const char *__restrict__ zero) {
  constexpr int VS_BYTES = NT * 16 * 128;             // V (then O) tile of one wave: [16*NT][64] f16
  __shared__ __attribute__((aligned(16))) char smem[4 * VS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t prob = (int64_t)blockIdx.x * 4 + wave;
  const bool live = prob < nproblems;
  if (!live) prob = nproblems - 1;                     // keep EXEC full for the transposed reads
  const int hd = (int)(prob % heads);
  const int64_t bp = prob / heads;                     // b*hw + p
  const int64_t b = bp / hw, p = bp - b * hw;
  const int64_t tok0 = b * frames * hw + p;            // row of frame 0
  char *vs = smem + wave * VS_BYTES;

  const int fr = lane & 15, fq = lane >> 4;
  // ---- V -> LDS (rows = frames, zero rows past `frames`)
  {
    const int lrow = lane >> 3, lchunk = lane & 7;
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) {
      const int f = i * 8 + lrow;
      const f16 *src = f < frames ? v + (tok0 + (int64_t)f * hw) * ldv + hd * 64 + lchunk * 8
                                  : (const f16 *)(zero + lchunk * 16);
      glds16(src, vs + i * 1024);
    }
  }
  // ---- Q, K fragments straight from global memory
  f16x8 qf[NT][2], kf[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int f = t * 16 + fr;
    const bool ok = f < frames;
    const f16 *qp = ok ? q + (tok0 + (int64_t)f * hw) * ldq + hd * 64 + 8 * fq : (const f16 *)zero;
    const f16 *kp = ok ? k + (tok0 + (int64_t)f * hw) * ldk + hd * 64 + 8 * fq : (const f16 *)zero;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      qf[t][s] = *(const f16x8 *)(qp + (ok ? 32 * s : 0));
      kf[t][s] = *(const f16x8 *)(kp + (ok ? 32 * s : 0));
    }
  }
  // ---- S^T[kt][qt]: col = query frame (fr) of tile qt, row = key 16kt + 4fq + e
  f32x4 sacc[NT][NT];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      sacc[kt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s)
        sacc[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt][s], qf[qt][s], sacc[kt][qt], 0, 0, 0);
    }
  // ---- softmax over keys for every query column, P normalised in place
  f16x4 pf[NT][NT];   // [qt][kt]
#pragma unroll
  for (int qt = 0; qt < NT; ++qt) {
    float mx = NEG_BIG;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int key = 16 * kt + 4 * fq + e;
        if (key >= frames) sacc[kt][qt][e] = NEG_BIG;
        mx = fmaxf(mx, sacc[kt][qt][e]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mb = mx * scale_log2e;
    float sum = 0.f;
    float pv[NT][4];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pv[kt][e] = fast_exp2(sacc[kt][qt][e] * scale_log2e - mb);
        sum += pv[kt][e];
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int e = 0; e < 4; ++e) pf[qt][kt][e] = (f16)(pv[kt][e] * inv);
  }
  // ---- O[qt][dt] = sum_kt P[qt][kt] . V[kt][dt]   (B operand by transposed LDS read)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32x4 oacc[NT][4];
  {
    const int q_ = fr >> 2, pp = fr & 3;
#pragma unroll
    for (int qt = 0; qt < NT; ++qt)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oacc[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const f16x4 vf = lds_tr16(vs + (16 * kt + 4 * fq + q_) * 128 + (16 * dt + 4 * pp) * 2);
#pragma unroll
        for (int qt = 0; qt < NT; ++qt)
          oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(pf[qt][kt], vf, oacc[qt][dt], 0, 0, 0);
      }
  }
  // ---- O -> LDS (reusing the V tile) -> 16-byte row stores
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int qt = 0; qt < NT; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        *(f16 *)(vs + (16 * qt + 4 * fq + e) * 128 + (16 * dt + fr) * 2) = (f16)oacc[qt][dt][e];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  {
    const int lrow = lane >> 3, lchunk = lane & 7;
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) {
      const int f = i * 8 + lrow;
      if (live && f < frames) {
        const f16x8 w = *(const f16x8 *)(vs + f * 128 + lchunk * 16);
        *(f16x8 *)(o + (tok0 + (int64_t)f * hw) * ldo + hd * 64 + lchunk * 8) = w;
      }
    }
  }
}

}  // namespace

int sp_attn_spatial_launch(const void *q, const void *k, const void *v, void *o, int64_t ldq, int64_t ldk, int64_t ldv,
                           int64_t ldo, int batch, int seq, int heads, float scale, const void *zero_page,
                           const unsigned *only_flagged, void *stream, const char *who) {
  SP_REQUIRE(q && k && v && o && zero_page, "%s: null pointer", who);
  SP_REQUIRE(batch > 0 && seq > 0 && heads > 0, "%s: batch/seq/heads must be positive", who);
  SP_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "%s: strides must be multiples of 8", who);
  SP_REQUIRE((int64_t)batch * heads <= 65535, "%s: batch*heads too large", who);
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(attn_spatial_kernel, dim3((seq + 127) / 128, batch * heads), dim3(256), 0,
                     (hipStream_t)stream, (const f16 *)q, (const f16 *)k, (const f16 *)v, (f16 *)o, ldq, ldk,
                     ldv, ldo, seq, heads, scale * 1.4426950408889634f, (const char *)zero_page, only_flagged);
  SP_CHECK_LAUNCH(who);
  return SP_OK;
}

extern "C" int sp_attn_spatial_f16(const void *q, const void *k, const void *v, void *o, int64_t ldq,
                                   int64_t ldk, int64_t ldv, int64_t ldo, int batch, int seq, int heads,
                                   float scale, const void *zero_page, void *stream) {
  return sp_attn_spatial_launch(q, k, v, o, ldq, ldk, ldv, ldo, batch, seq, heads, scale, zero_page, nullptr, stream,
                                "sp_attn_spatial_f16");
}

extern "C" int sp_attn_temporal_f16(const void *q, const void *k, const void *v, void *o, int64_t ldq,
                                    int64_t ldk, int64_t ldv, int64_t ldo, int batch, int frames, int64_t hw,
                                    int heads, float scale, const void *zero_page, void *stream) {
  SP_REQUIRE(q && k && v && o && zero_page, "sp_attn_temporal_f16: null pointer");
  SP_REQUIRE(batch > 0 && frames > 0 && frames <= 32 && hw > 0 && heads > 0,
             "sp_attn_temporal_f16: unsupported shape (frames=%d must be in [1,32])", frames);
  SP_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "sp_attn_temporal_f16: strides must be multiples of 8");
  const int64_t nprob = (int64_t)batch * hw * heads;
  const unsigned grid = (unsigned)((nprob + 3) / 4);
  const float sl = scale * 1.4426950408889634f;
  hipStream_t s = (hipStream_t)stream;
  SP_CLEAR_STALE_ERROR();
  if (frames <= 16)
    hipLaunchKernelGGL(attn_temporal_kernel<1>, dim3(grid), dim3(256), 0, s, (const f16 *)q, (const f16 *)k,
                       (const f16 *)v, (f16 *)o, ldq, ldk, ldv, ldo, frames, hw, heads, nprob, sl,
                       (const char *)zero_page);
  else
    hipLaunchKernelGGL(attn_temporal_kernel<2>, dim3(grid), dim3(256), 0, s, (const f16 *)q, (const f16 *)k,
                       (const f16 *)v, (f16 *)o, ldq, ldk, ldv, ldo, frames, hw, heads, nprob, sl,
                       (const char *)zero_page);
  SP_CHECK_LAUNCH("sp_attn_temporal_f16");
  return SP_OK;
}

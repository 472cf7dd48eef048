// EXPERIMENT, NOT BUILT INTO libsvdpipe_hip.so (round 3; DESIGN.md section 3 "Spatial attention, round 3").
// Measured against attn_spatial_kernel (three waves per SIMD) at 14 x 9,216 x 5 heads, same process, results
// bit-identical: QB = 4 0.82x, QB = 3 0.85-0.87x (hipcc -mllvm -amdgpu-mfma-vgpr-form=1; without that flag the scores
// land in AGPRs and every query block and tile pays 64 v_accvgpr moves).  A wave alone on its SIMD exposes every LDS
// round trip and every MFMA -> VALU dependency that two partner waves hide for free; the compiler-scheduled stream
// (sched_group_barrier pins only part of the interleave across the rescale branches) leaves ~45 % of the issue slots
// empty.  Kept as the starting point for a hand-placed stream; to try it again: add it to csrc/Makefile's SRCS with
// the flag above and route sp_attn_spatial_f16 to sp_attn_wide_launch for seq % (128*QB) == 0.
//
// Spatial self-attention for long rows (head_dim 64, fp16 in / fp32 accumulate, no mask): ONE wave per SIMD with the
// whole 512-register file, QB blocks of 32 query rows per wave (workgroup = 4 waves x QB x 32 rows).
//
// Why: attention.hip's kernel (three waves per SIMD, 32 query rows each) is bound by vector ISSUE, and a third of what
// a wave issues per K/V tile does not scale with its query rows: the LDS-DMA pieces of the tile (4 per wave), the 16
// transposed V reads, the barrier and the loop bookkeeping.  Here a wave takes QB query blocks through every tile, so
// those are paid once per QB*32 rows, the V fragments are read once and used QB times, and the matrix pipe is kept
// fed from WITHIN the wave: the instruction stream is software-pipelined over the query blocks,
//
//     step j:   softmax(block j)  [vector]   beside   O += V.P(block j-1),  S(block j+1) = K.Q - m  [matrix]
//
// with the MFMAs placed between the vector instructions in program order (an in-order wave only overlaps an MFMA with
// the vector work that FOLLOWS it; __builtin_amdgcn_sched_group_barrier pins the interleave).  Same arithmetic as
// attn_spatial_kernel<1>: transposed scores (a lane holds 32 keys of one query column), scores start from -m, exp2 on
// pre-scaled Q, deferred rescale (threshold 2^8), row sums on the matrix pipe, P as accumulator-as-operand.
// Host contract: seq a multiple of 128*QB (no ragged tiles, no partial query blocks); everything else goes to
// attention.hip.
#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// V: experiment switches.  bit 0: no sched_group_barrier hints; bit 1 (TIMING ONLY): no rescale after the first tile;
// bit 2 (TIMING ONLY): no row maximum after the first tile either.
template <int QB, int V = 0>
__global__ __launch_bounds__(256, 1) void attn_spatial_wide_kernel(
    const f16 *__restrict__ q, const f16 *__restrict__ k, const f16 *__restrict__ v, f16 *__restrict__ o,
    int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int seq, int heads, float scale_log2e) {
  constexpr int KV = 64;
  constexpr int K_BYTES = KV * 128, STAGE = 2 * K_BYTES, RING = 3;
  __shared__ __attribute__((aligned(16))) char smem[RING * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.y;
  const int b = bh / heads, hd = bh - b * heads;
  const int r = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)b * seq;
  const int q0 = blockIdx.x * (128 * QB) + wave * (32 * QB);

  // Q fragments (B operand of S^T = K.Q^T), pre-multiplied by scale*log2(e) (fp32 product, one rounding to fp16)
  f16x8 qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const f16 *qp = q + (row0 + q0 + qb * 32 + r) * ldq + hd * 64 + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f16x8 raw = *(const f16x8 *)(qp + 16 * s);
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[qb][s][e] = (f16)((float)raw[e] * scale_log2e);
    }
  }

  // K/V staging as in attention.hip: each wave moves 16 rows of K and of V per tile (2 x 2 LDS-DMA pieces of 8 rows)
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
  auto stage = [&](int buf) {
    char *sk = smem + buf * STAGE;
    char *sv = sk + K_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      glds16(kb + kofl[i], sk + (wave * 16 + i * 8) * 128);
      glds16(vb + vofl[i], sv + (wave * 16 + i * 8) * 128);
    }
    kb += kstep;
    vb += vstep;
  };

  // operand read offsets (see attention.hip)
  int koff[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) koff[kt] = (kt * 32 + r) * 128;
  const int kswz = (r >> 1) & 7;
  const int g = lane >> 4, i16 = lane & 15, q_ = i16 >> 2, pp = i16 & 3;
  const int vrow_l = 4 * (g >> 1) + q_;
  const int vchunk_l = 2 * (g & 1) + (pp >> 1);
  const int vswz = (q_ >> 1) << 2;
  const int vbyte_l = (pp & 1) * 8;
  const int vlane0 = vrow_l * 128 + ((vchunk_l ^ vswz) << 4) + vbyte_l;
  const int vlane1 = vrow_l * 128 + (((4 + vchunk_l) ^ vswz) << 4) + vbyte_l;

  f32x16 oacc[QB][2], negm[QB];
  f32x4 lacc[QB];
  float m_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = 0.f;
    lacc[qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 16; ++e) { oacc[qb][0][e] = 0.f; oacc[qb][1][e] = 0.f; negm[qb][e] = 0.f; }
  }
  // selector operand of the row-sum MFMA (attention.hip): element 0 of the 16x16 accumulator = this lane's own row sum
  f16x8 lsel;
  {
    const int m16 = lane & 15, kb4 = lane >> 4;
    const bool one = ((m16 & 7) == 0 && (kb4 & 1) == 0) || ((m16 & 7) == 4 && (kb4 & 1) == 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) lsel[e] = one ? (f16)1.f : (f16)0.f;
  }

  const int ntiles = seq / KV;
  // 3-deep K/V ring: tiles t+1 and t+2 are in flight while tile t is consumed
  stage(0);
  if (ntiles > 1) stage(1);
  if (ntiles > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- the three pieces of a query block's tile, written so that they can be placed beside each other
  // S^T = K.Q^T - m for block qb (8 MFMAs, K fragments from LDS)
  auto qk = [&](int qb, const char *sk, f32x16 (&sacc)[2]) {
    f16x8 kf[2][4];                           // all eight fragments requested ahead of the first MFMA: a wave alone on
#pragma unroll                                // its SIMD has nobody to cover an LDS round trip in front of each MFMA
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s = 0; s < 4; ++s) kf[kt][s] = *(const f16x8 *)(sk + koff[kt] + (((2 * s + h) ^ kswz) << 4));
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kt][s], qf[qb][s], s == 0 ? negm[qb] : sacc[kt], 0, 0, 0);
  };
  // O^T += V^T.P^T and the row sums for block qb (8 + 4 MFMAs, V fragments in registers)
  auto pv = [&](int qb, const u32x4 (&pw)[2][2], const u32x2 (&vr)[2][8]) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const u32x4 vw = {vr[dt][4 * kt + 2 * s][0], vr[dt][4 * kt + 2 * s][1], vr[dt][4 * kt + 2 * s + 1][0],
                            vr[dt][4 * kt + 2 * s + 1][1]};
          oacc[qb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, vw),
                                                               __builtin_bit_cast(f16x8, pw[kt][s]), oacc[qb][dt], 0, 0, 0);
          if (dt == 0)
            lacc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lsel, __builtin_bit_cast(f16x8, pw[kt][s]), lacc[qb], 0, 0, 0);
        }
  };

  int buf = 0;
  for (int t = 0; t < ntiles; ++t) {
    if (t + 2 < ntiles) stage(buf >= 1 ? buf - 1 : RING - 1);       // slot of tile t-1 = (t+2) % 3
    const char *sk = smem + buf * STAGE;
    const char *sv = sk + K_BYTES;

    f32x16 sacc[2][2];                       // scores of the block in its softmax / of the next block
    u32x4 pw[2][2][2];                       // packed probabilities of the block in its softmax / of the previous block
    qk(0, sk, sacc[0]);
    // V fragments of the whole tile, read once for all QB blocks (inline asm: for the builtin hipcc waits for the
    // LDS-DMA in flight); issued behind the first score MFMAs, which cover the read latency
    u32x2 vr[2][8];
    {
      const unsigned va[2] = {(unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)(sv + vlane0),
                              (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)(sv + vlane1)};
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vr[dt][i]) : "v"(va[dt]), "n"(i * 8 * 128) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(vr[0][0]), "+v"(vr[0][1]), "+v"(vr[0][2]), "+v"(vr[0][3]), "+v"(vr[0][4]), "+v"(vr[0][5]),
                     "+v"(vr[0][6]), "+v"(vr[0][7]), "+v"(vr[1][0]), "+v"(vr[1][1]), "+v"(vr[1][2]), "+v"(vr[1][3]),
                     "+v"(vr[1][4]), "+v"(vr[1][5]), "+v"(vr[1][6]), "+v"(vr[1][7])::"memory");
    }

#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      f32x16(&sc)[2] = sacc[qb & 1];
      // ---- (a) row maximum of block qb   beside   PV of block qb-1
      // (V & 16: the maximum FIRST, the MFMAs behind it, so that the rescale decision is old news when the branch comes)
      if (qb > 0 && !(V & 16)) pv(qb - 1, pw[(qb - 1) & 1], vr);
      float mt = 0.f;
      if (!(V & 4) || t == 0) {
        if (V & 8) {
          // tree: a wave alone on its SIMD pays the latency of every dependent max; four levels instead of sixteen
          float a[11];
#pragma unroll
          for (int i = 0; i < 10; ++i) {
            const int e = 3 * i;
            a[i] = fmaxf(fmaxf(sc[e >> 4][e & 15], sc[(e + 1) >> 4][(e + 1) & 15]), sc[(e + 2) >> 4][(e + 2) & 15]);
          }
          a[10] = fmaxf(sc[1][14], sc[1][15]);
          const float b0 = fmaxf(fmaxf(a[0], a[1]), a[2]), b1 = fmaxf(fmaxf(a[3], a[4]), a[5]);
          const float b2 = fmaxf(fmaxf(a[6], a[7]), a[8]), b3 = fmaxf(a[9], a[10]);
          mt = fmaxf(fmaxf(fmaxf(b0, b1), b2), b3);
        } else {
          mt = fmaxf(fmaxf(sc[0][0], sc[0][1]), sc[0][2]);
#pragma unroll
          for (int e = 3; e < 31; e += 2) mt = fmaxf(fmaxf(mt, sc[e >> 4][e & 15]), sc[(e + 1) >> 4][(e + 1) & 15]);
          mt = fmaxf(mt, sc[1][15]);
        }
        float ma = mt, mb = mt;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
        mt = fmaxf(ma, mb);
      }
      if (qb > 0 && (V & 16)) {
        __builtin_amdgcn_sched_barrier(0);
        pv(qb - 1, pw[(qb - 1) & 1], vr);
      }
      if (qb > 0 && !(V & 1) && !(V & 16)) {          // 12 MFMAs beside ~24 vector instructions
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     // two VALU
        }
      }
      constexpr float RESCALE_LOG2 = 8.0f;
      const float delta = t == 0 ? mt : (mt > RESCALE_LOG2 ? mt : 0.f);
      if (((V & 6) == 0 || t == 0) && __builtin_amdgcn_ballot_w64(delta != 0.f) != 0) {       // rare after the first tile
        const float alpha = t == 0 ? 1.f : fast_exp2(-delta);
        lacc[qb][0] *= alpha;
#pragma unroll
        for (int e = 0; e < 16; ++e) { oacc[qb][0][e] *= alpha; oacc[qb][1][e] *= alpha; }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int e = 0; e < 16; ++e) sc[kt][e] -= delta;
        m_run[qb] += delta;
#pragma unroll
        for (int e = 0; e < 16; ++e) negm[qb][e] = -m_run[qb];
      }
      // ---- (b) exponentials of block qb   beside   the scores of block qb+1
      if (qb + 1 < QB) qk(qb + 1, sk, sacc[(qb + 1) & 1]);
      u32x4(&pc)[2][2] = pw[qb & 1];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          const f32x2 p = {fast_exp2(sc[kt][e]), fast_exp2(sc[kt][e + 1])};
          pc[kt][e >> 3][(e & 7) >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(p, f16x2));
        }
      if (qb + 1 < QB && !(V & 1)) {                  // 8 MFMAs (+ their 8 K reads) beside 48 vector instructions
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);       // the eight K fragments first
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);       // a first run of exponentials covers their latency
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);     // five VALU (exp, exp, convert, ...)
        }
      }
    }
    pv(QB - 1, pw[(QB - 1) & 1], vr);
    // tile t+1 landed (this wave's pieces; tile t+2's four may stay in flight); every wave is done reading tile t's slot
    if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    buf = buf + 1 == RING ? 0 : buf + 1;
  }

  // ---- finalize: O^T[d][q] / l ; lane holds d = 32dt + (e&3) + 8(e>>2) + 4h for query q0 + 32 qb + r
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float inv = 1.0f / lacc[qb][0];
    f16 *op = o + (row0 + q0 + qb * 32 + r) * ldo + hd * 64 + 4 * h;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (f16)(oacc[qb][dt][4 * c + e] * inv);
        *(f16x4 *)(op + 32 * dt + 8 * c) = w;
      }
  }
}

}  // namespace

// called by sp_attn_spatial_f16 (attention.hip) for rows it supports; returns false when the shape is not its own
bool sp_attn_wide_supported(int seq, int qb) { return seq >= 4096 && seq % (128 * qb) == 0; }

int sp_attn_wide_launch(const void *q, const void *k, const void *v, void *o, int64_t ldq, int64_t ldk, int64_t ldv,
                        int64_t ldo, int batch, int seq, int heads, float scale_log2e, int qb, hipStream_t stream) {
  SP_CLEAR_STALE_ERROR();
  if (qb == 4)
    hipLaunchKernelGGL(attn_spatial_wide_kernel<4>, dim3(seq / 512, batch * heads), dim3(256), 0, stream, (const f16 *)q,
                       (const f16 *)k, (const f16 *)v, (f16 *)o, ldq, ldk, ldv, ldo, seq, heads, scale_log2e);
  else if (qb == 3)
    hipLaunchKernelGGL(attn_spatial_wide_kernel<3>, dim3(seq / 384, batch * heads), dim3(256), 0, stream, (const f16 *)q,
                       (const f16 *)k, (const f16 *)v, (f16 *)o, ldq, ldk, ldv, ldo, seq, heads, scale_log2e);
  else
    hipLaunchKernelGGL(attn_spatial_wide_kernel<2>, dim3(seq / 256, batch * heads), dim3(256), 0, stream, (const f16 *)q,
                       (const f16 *)k, (const f16 *)v, (f16 *)o, ldq, ldk, ldv, ldo, seq, heads, scale_log2e);
  SP_CHECK_LAUNCH("sp_attn_spatial_f16(wide)");
  return SP_OK;
}

// experiments harness (linked into libsvdpipe_hip_exp.so by hand: see tools/bench_attn_wide.py)
extern "C" int sp_exp_attn_wide(const void *q, const void *k, const void *v, void *o, int64_t ldq, int64_t ldk, int64_t ldv,
                                int64_t ldo, int batch, int seq, int heads, float scale, int qb, int variant, void *stream) {
  const float sl = scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  if (seq % (128 * qb)) return -1;
  SP_CLEAR_STALE_ERROR();
#define WIDE(QBV, VV)                                                                                                     \
  hipLaunchKernelGGL((attn_spatial_wide_kernel<QBV, VV>), dim3(seq / (128 * QBV), batch * heads), dim3(256), 0, st,        \
                     (const f16 *)q, (const f16 *)k, (const f16 *)v, (f16 *)o, ldq, ldk, ldv, ldo, seq, heads, sl)
  if (qb == 3) { switch (variant) { case 1: WIDE(3, 1); break; case 7: WIDE(3, 7); break; case 9: WIDE(3, 9); break;
                                    case 11: WIDE(3, 11); break; case 8: WIDE(3, 8); break; case 24: WIDE(3, 24); break; case 25: WIDE(3, 25); break; case 10: WIDE(3, 10); break; case 6: WIDE(3, 6); break; default: return -2; } }
  else if (qb == 4) { switch (variant) { case 1: WIDE(4, 1); break; case 9: WIDE(4, 9); break; case 8: WIDE(4, 8); break;
                                         case 11: WIDE(4, 11); break; default: return -2; } }
  else if (qb == 2) { switch (variant) { case 1: WIDE(2, 1); break; case 7: WIDE(2, 7); break; case 9: WIDE(2, 9); break;
                                         case 11: WIDE(2, 11); break; case 8: WIDE(2, 8); break; case 6: WIDE(2, 6); break; case 0: WIDE(2, 0); break; case 10: WIDE(2, 10); break; case 16: WIDE(2, 16); break; case 24: WIDE(2, 24); break; case 25: WIDE(2, 25); break; case 17: WIDE(2, 17); break; default: return -2; } }
  else return -3;
#undef WIDE
  SP_CHECK_LAUNCH("sp_exp_attn_wide");
  return SP_OK;
}

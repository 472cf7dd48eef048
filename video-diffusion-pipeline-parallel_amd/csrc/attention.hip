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
//
// attn_temporal_kernel – sequences run across frames (14 or 25 tokens) for each pixel/head.
//   One wave per (pixel, head); Q/K fragments are loaded straight from global memory in MFMA operand
//   layout (token (f,p) = row f*hw+p, so the (B*HW, F, C) permute of the reference never exists),
//   S^T via v_mfma_f32_16x16x32_f16, softmax across the 4 lane groups, P (still in accumulator
//   layout) is directly the A operand of v_mfma_f32_16x16x16_f16 for O = P.V.  HBM-bound.
#include "common.h"

namespace {

typedef short v4i16 __attribute__((__vector_size__(8)));

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
    const char *__restrict__ zero) {
  constexpr int KV = 64;
  constexpr int K_BYTES = KV * 128, STAGE = 2 * K_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.y;
  const int b = bh / heads, hd = bh - b * heads;
  const int r = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)b * seq;
  const int q0 = blockIdx.x * 128 + wave * 32;

  // Q fragments (B operand of S^T = K.Q^T): lane holds Q[q0 + r][16s + 8h .. +8]
  f16x8 qf[4];
  {
    const bool ok = q0 + r < seq;
    const f16 *qp = ok ? q + (row0 + q0 + r) * ldq + hd * 64 + 8 * h : (const f16 *)zero;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const f16x8 *)(qp + (ok ? 16 * s : 0));
  }

  // K/V staging: each wave moves 16 rows of K and of V per tile (2 x 2 LDS-DMA pieces of 8 rows)
  const int lrow = lane >> 3, lchunk = lane & 7;
  const f16 *kbase = k + row0 * ldk + hd * 64;
  const f16 *vbase = v + row0 * ldv + hd * 64;
  auto stage = [&](int tile, int buf) {
    char *sk = smem + buf * STAGE;
    char *sv = sk + K_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wave * 16 + i * 8 + lrow;
      const int key = tile * KV + row;
      const bool ok = key < seq;
      const f16 *ks = ok ? kbase + (int64_t)key * ldk + ((lchunk ^ ((row >> 1) & 7)) << 3)
                         : (const f16 *)(zero + lchunk * 16);
      const f16 *vs = ok ? vbase + (int64_t)key * ldv + ((lchunk ^ (((row >> 1) & 1) << 2)) << 3)
                         : (const f16 *)(zero + lchunk * 16);
      glds16(ks, sk + (wave * 16 + i * 8) * 128);
      glds16(vs, sv + (wave * 16 + i * 8) * 128);
    }
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
  float m_run = NEG_BIG, l_run = 0.f;

  const int ntiles = (seq + KV - 1) / KV;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) stage(t + 1, buf ^ 1);
    const char *sk = smem + buf * STAGE;
    const char *sv = sk + K_BYTES;
    const char *sv0 = sv + vlane0, *sv1 = sv + vlane1;

    // ---- S^T = K.Q^T : two 32-key sub-tiles
    f32x16 sacc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[kt][e] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f16x8 kf = *(const f16x8 *)(sk + koff[kt] + (((2 * s + h) ^ kswz) << 4));
        sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc[kt], 0, 0, 0);
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
    float mt = sacc[0][0];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) mt = fmaxf(mt, sacc[kt][e]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
      const float alpha = fast_exp2((m_run - m_new) * scale_log2e);
      l_run *= alpha;
#pragma unroll
      for (int e = 0; e < 16; ++e) { oacc[0][e] *= alpha; oacc[1][e] *= alpha; }
      m_run = m_new;
    }
    const float mb = m_run * scale_log2e;
    // P -> fp16 pairwise (v_cvt_pk_f16_f32, round-to-nearest: one instruction per pair instead of two converts
    // and a pack)
    f16x8 pf[2][2];
    float lsum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const float p0 = fast_exp2(sacc[kt][e] * scale_log2e - mb);
        const float p1 = fast_exp2(sacc[kt][e + 1] * scale_log2e - mb);
        lsum += p0 + p1;
        const f16x2 pk = __builtin_convertvector((f32x2){p0, p1}, f16x2);
        pf[kt][e >> 3][e & 7] = pk[0];
        pf[kt][e >> 3][(e & 7) + 1] = pk[1];
      }
    l_run += lsum;

    // ---- O^T += V^T . P^T
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          f16x8 vf;
#pragma unroll
          for (int e2 = 0; e2 < 2; ++e2) {
            const f16x4 x = lds_tr16((dt ? sv1 : sv0) + (kt * 32 + 16 * s + 8 * e2) * 128);
            vf[4 * e2 + 0] = x[0]; vf[4 * e2 + 1] = x[1]; vf[4 * e2 + 2] = x[2]; vf[4 * e2 + 3] = x[3];
          }
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kt][s], oacc[dt], 0, 0, 0);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    buf ^= 1;
  }

  // ---- finalize: O^T[d][q] / l ; lane holds d = 32dt + (e&3) + 8(e>>2) + 4h for query q0 + r
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
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

extern "C" int sp_attn_spatial_f16(const void *q, const void *k, const void *v, void *o, int64_t ldq,
                                   int64_t ldk, int64_t ldv, int64_t ldo, int batch, int seq, int heads,
                                   float scale, const void *zero_page, void *stream) {
  SP_REQUIRE(q && k && v && o && zero_page, "sp_attn_spatial_f16: null pointer");
  SP_REQUIRE(batch > 0 && seq > 0 && heads > 0, "sp_attn_spatial_f16: batch/seq/heads must be positive");
  SP_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "sp_attn_spatial_f16: strides must be multiples of 8");
  SP_REQUIRE((int64_t)batch * heads <= 65535, "sp_attn_spatial_f16: batch*heads too large");
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(attn_spatial_kernel, dim3((seq + 127) / 128, batch * heads), dim3(256), 0,
                     (hipStream_t)stream, (const f16 *)q, (const f16 *)k, (const f16 *)v, (f16 *)o, ldq, ldk,
                     ldv, ldo, seq, heads, scale * 1.4426950408889634f, (const char *)zero_page);
  SP_CHECK_LAUNCH("sp_attn_spatial_f16");
  return SP_OK;
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

// GroupNorm(+SiLU) and LayerNorm for channels-last fp16 token matrices (HBM-bound kernels).
//
// GroupNorm: three launches.
//   stats : grid (instances, splits); every thread owns one 8-channel octet (16-byte loads, rows
//           strided by P), accumulates per-channel sum / sum-of-squares of (x - ref) in fp32 registers, the block
//           reduces them in LDS in a FIXED order (deterministic, no atomics) to per-group partials
//           and writes [instance][split][group][2].  ref = the group's first element of the instance's first row:
//           sums of shifted values do not cancel when |mean| >> std (var = E[d^2] - E[d]^2 with E[d] ~ std).
//   final : one block per instance folds the partials (fixed order, fp64) into mean / rstd.
//   apply : streams rows: y = x*scale[c] + shift[c], optional SiLU, 16-byte loads/stores.
// The second read of x is served mostly by the 256 MiB Infinity Cache (tensors are <= 83 MB).
#include "common.h"

namespace {

__host__ __device__ inline int gn_rows_per_iter(int oc) { int p = 512 / oc; return p < 1 ? 1 : p; }

__global__ void gn_stats_kernel(const f16 *__restrict__ x, float *__restrict__ part, int64_t rows,
                                int c, int groups, int splits, int64_t per, int64_t ldx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *sh = (float *)smem;                      // [P][C][2], then [parts][groups][2] behind it
  const int oc = c >> 3;
  const int P = gn_rows_per_iter(oc);
  const int tid = threadIdx.x;
  const int o = tid % oc, pr = tid / oc;
  const int inst = blockIdx.x, split = blockIdx.y;
  const int64_t r0 = (int64_t)split * per;        // per: a multiple of 4*P (host), so every block but the last runs
  int64_t r1 = r0 + per; if (r1 > rows) r1 = rows;   // whole batches of four rows per thread
  float s[8], ss[8], ref[8];
  const int cpg = c / groups;
  {
    // the group's first element of the instance's first row, for each of this thread's 8 channels
    const f16 *row0 = x + ((int64_t)inst * rows) * ldx;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s[e] = ss[e] = 0.f;
      ref[e] = o < oc ? (float)row0[((o * 8 + e) / cpg) * cpg] : 0.f;
    }
  }
  const f16 *base = x + ((int64_t)inst * rows) * ldx + o * 8;
  if (pr < P) {
    // batches of four independent 16-byte loads per thread, double-buffered: the next batch is in flight while this one
    // is accumulated (a block has only ~10 row-iterations: without the overlap it is a chain of exposed round trips)
    auto acc4 = [&](const f16x8 (&v)[4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)v[u][e] - ref[e]; s[e] += f; ss[e] += f * f; }
    };
    int64_t r = r0 + pr;
    const int64_t step = 4 * (int64_t)P;
    if (r + 3 * (int64_t)P < r1) {
      f16x8 va[4], vb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) va[u] = *(const f16x8 *)(base + (r + (int64_t)u * P) * ldx);
      r += step;
      for (; r + 3 * (int64_t)P < r1; r += step) {
#pragma unroll
        for (int u = 0; u < 4; ++u) vb[u] = *(const f16x8 *)(base + (r + (int64_t)u * P) * ldx);
        acc4(va);
#pragma unroll
        for (int u = 0; u < 4; ++u) va[u] = vb[u];
      }
      acc4(va);
    }
    for (; r < r1; r += P) {
      const f16x8 v = *(const f16x8 *)(base + r * ldx);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float f = (float)v[e] - ref[e]; s[e] += f; ss[e] += f * f; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sh[((pr * c) + o * 8 + e) * 2 + 0] = s[e];
      sh[((pr * c) + o * 8 + e) * 2 + 1] = ss[e];
    }
  }
  __syncthreads();
  // fold [P][cpg] values per group in a FIXED order: `parts` threads per group take every parts-th element, then one
  // thread per group adds the parts in order (deterministic, no atomics)
  const int parts = (int)blockDim.x / groups;        // >= 1
  float *sh2 = sh + (size_t)P * c * 2;
  {
    const int g = tid % groups, pt = tid / groups;
    if (pt < parts) {
      float a = 0.f, b = 0.f;
      const int n = P * cpg;
      for (int i = pt; i < n; i += parts) {
        const int q = i / cpg, ch = g * cpg + (i - q * cpg);
        a += sh[(q * c + ch) * 2 + 0];
        b += sh[(q * c + ch) * 2 + 1];
      }
      sh2[(pt * groups + g) * 2 + 0] = a;
      sh2[(pt * groups + g) * 2 + 1] = b;
    }
  }
  __syncthreads();
  if (tid < groups) {
    float a = 0.f, b = 0.f;
    for (int q = 0; q < parts; ++q) { a += sh2[(q * groups + tid) * 2]; b += sh2[(q * groups + tid) * 2 + 1]; }
    float *dst = part + (((int64_t)inst * splits + split) * groups + tid) * 2;
    dst[0] = a; dst[1] = b;
  }
}

// one block per instance: fold the per-split partials into mean / rstd (fixed order, fp64)
__global__ __launch_bounds__(1024) void gn_finalize_kernel(const f16 *__restrict__ x, const float *__restrict__ part,
                                                           float *__restrict__ stats, int64_t rows, int c,
                                                           int groups, int splits, float eps, int64_t ldx) {
  __shared__ double sh[1024 * 2];
  const int tid = threadIdx.x, inst = blockIdx.x;
  const int slices = 1024 / groups;
  const int g = tid % groups, sl = tid / groups;
  double a = 0.0, b = 0.0;
  if (sl < slices) {
    const float2 *src = (const float2 *)(part + ((int64_t)inst * splits * groups + g) * 2);
    int sp = sl;
    for (; sp + 3 * slices < splits; sp += 4 * slices) {       // four loads in flight
      float2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = src[(int64_t)(sp + u * slices) * groups];
#pragma unroll
      for (int u = 0; u < 4; ++u) { a += v[u].x; b += v[u].y; }
    }
    for (; sp < splits; sp += slices) { const float2 v = src[(int64_t)sp * groups]; a += v.x; b += v.y; }
  }
  sh[tid * 2] = a; sh[tid * 2 + 1] = b;
  __syncthreads();
  if (tid < groups) {
    a = 0.0; b = 0.0;
    for (int q = 0; q < slices; ++q) { a += sh[(q * groups + tid) * 2]; b += sh[(q * groups + tid) * 2 + 1]; }
    const int cpg = c / groups;
    const double cnt = (double)rows * cpg;
    const double dmean = a / cnt;                        // mean of (x - ref): of the order of the standard deviation
    double var = b / cnt - dmean * dmean; if (var < 0.0) var = 0.0;
    const double ref = (double)(float)x[((int64_t)inst * rows) * ldx + tid * cpg];
    stats[((int64_t)inst * groups + tid) * 2] = (float)(ref + dmean);
    stats[((int64_t)inst * groups + tid) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// `part` != null: the block folds the per-split partials of its instance itself (fixed order, fp64: every block of an
// instance computes bit-identical statistics), so that no finalize launch sits between the two passes; used where an
// instance has few splits (the per-frame norms), see sp_groupnorm_f16.
__global__ void gn_apply_kernel(const f16 *__restrict__ x, const float *__restrict__ stats,
                                const float *__restrict__ part, int splits, float eps,
                                const float *__restrict__ gamma, const float *__restrict__ beta,
                                f16 *__restrict__ y, int64_t rows, int c, int groups, int silu,
                                int64_t rows_per_block, int64_t ldx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int oc = c >> 3;
  const int P = gn_rows_per_iter(oc);
  const int tid = threadIdx.x;
  const int inst = blockIdx.x;
  const int cpg = c / groups;
  const float *st = stats + (int64_t)inst * groups * 2;
  if (part) {
    double *red = (double *)smem;                   // [slices][groups][2]
    float *stl = (float *)(red + (size_t)((int)blockDim.x / groups) * groups * 2);
    const int slices = (int)blockDim.x / groups;
    const int g = tid % groups, sl = tid / groups;
    if (sl < slices) {
      const float2 *src = (const float2 *)(part + ((int64_t)inst * splits * groups + g) * 2);
      double a = 0.0, b = 0.0;
      int sp = sl;
      for (; sp + 3 * slices < splits; sp += 4 * slices) {       // four loads in flight
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = src[(int64_t)(sp + u * slices) * groups];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a += v[u].x; b += v[u].y; }
      }
      for (; sp < splits; sp += slices) { const float2 v = src[(int64_t)sp * groups]; a += v.x; b += v.y; }
      red[(sl * groups + g) * 2] = a; red[(sl * groups + g) * 2 + 1] = b;
    }
    __syncthreads();
    if (tid < groups) {
      double a = 0.0, b = 0.0;
      for (int q = 0; q < slices; ++q) { a += red[(q * groups + tid) * 2]; b += red[(q * groups + tid) * 2 + 1]; }
      const double cnt = (double)rows * cpg;
      const double dmean = a / cnt;
      double var = b / cnt - dmean * dmean; if (var < 0.0) var = 0.0;
      const double ref = (double)(float)x[((int64_t)inst * rows) * ldx + tid * cpg];
      stl[tid * 2] = (float)(ref + dmean);
      stl[tid * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    st = stl;
  }
  const int o = tid % oc, pr = tid / oc;
  if (pr >= P) return;
  float sc[8], sf[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = o * 8 + e;
    const int g = ch / cpg;
    const float ga = gamma ? gamma[ch] : 1.f, be = beta ? beta[ch] : 0.f;
    sc[e] = st[g * 2 + 1] * ga;
    sf[e] = be - st[g * 2] * sc[e];
  }
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  int64_t r1 = r0 + rows_per_block; if (r1 > rows) r1 = rows;
  const f16 *xb = x + ((int64_t)inst * rows) * ldx + o * 8;
  f16 *yb = y + ((int64_t)inst * rows) * c + o * 8;
  auto emit4 = [&](const f16x8 (&v)[4], int64_t rr) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f16x8 w;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = (float)v[u][e] * sc[e] + sf[e];
        if (silu) f = silu_f(f);
        w[e] = (f16)f;
      }
      *(f16x8 *)(yb + (rr + (int64_t)u * P) * c) = w;
    }
  };
  int64_t r = r0 + pr;
  const int64_t step = 4 * (int64_t)P;
  if (r + 3 * (int64_t)P < r1) {       // batches of four loads per thread, the next batch in flight while this one is written
    f16x8 va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) va[u] = *(const f16x8 *)(xb + (r + (int64_t)u * P) * ldx);
    int64_t rp = r;
    r += step;
    for (; r + 3 * (int64_t)P < r1; r += step) {
#pragma unroll
      for (int u = 0; u < 4; ++u) vb[u] = *(const f16x8 *)(xb + (r + (int64_t)u * P) * ldx);
      emit4(va, rp);
      rp = r;
#pragma unroll
      for (int u = 0; u < 4; ++u) va[u] = vb[u];
    }
    emit4(va, rp);
  }
  for (; r < r1; r += P) {
    const f16x8 v = *(const f16x8 *)(xb + r * ldx);
    f16x8 w;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f = (float)v[e] * sc[e] + sf[e];
      if (silu) f = silu_f(f);
      w[e] = (f16)f;
    }
    *(f16x8 *)(yb + r * c) = w;
  }
}

// Single-launch GroupNorm for slabs that fit in registers: one workgroup per (instance, group) loads its
// rows x (C/groups) slab once (VEC halves per access), reduces sum / sum-of-squares in a fixed order (wave shuffles,
// then one thread folds the per-wave partials: deterministic), normalises from registers and writes.  One read and one
// write instead of two reads and one write, and one launch instead of three: the 576- and 144-token levels were
// launch-bound (30 us for a 20 MB tensor).
template <int VEC, int MAXIT>
__global__ __launch_bounds__(512) void gn_fused_kernel(const f16 *__restrict__ x, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, f16 *__restrict__ y,
                                                       int64_t rows, int c, int groups, float eps, int silu,
                                                       int64_t ldx) {
  typedef _Float16 hv __attribute__((ext_vector_type(VEC)));
  __shared__ float red[2 * 8 + 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int inst = blockIdx.x / groups, g = blockIdx.x - inst * groups;
  const int cpg = c / groups, vpr = cpg / VEC;               // vectors per row
  const f16 *xb = x + (int64_t)inst * rows * ldx + (int64_t)g * cpg;
  f16 *yb = y + (int64_t)inst * rows * c + (int64_t)g * cpg;
  // thread -> (vector q of the row, row rr of every pass of RP rows): one division per thread, 32-bit addressing
  const int rp = 512 / vpr;                                  // rows per pass
  const int q = tid % vpr, rr = tid / vpr;
  const bool act = rr < rp;
  const int nrows = (int)rows;
  hv v[MAXIT];
  bool live[MAXIT];
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int r = rr + it * rp;
    live[it] = act && r < nrows;
    if (live[it]) {
      v[it] = *(const hv *)(xb + (int64_t)r * ldx + q * VEC);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[it][e] = (f16)0.f;
    }
  }
  // two passes over the registers: the mean first, then the sum of squared deviations (no E[x^2] - mean^2 cancellation)
#pragma unroll
  for (int it = 0; it < MAXIT; ++it)
#pragma unroll
    for (int e = 0; e < VEC; ++e) s += (float)v[it][e];
  // keep the slab PACKED across the reductions (otherwise the compiler keeps the fp32 copies alive: 2x the registers)
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) asm volatile("" : "+v"(v[it]));
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) {
    double a = 0.0;
    for (int w = 0; w < 8; ++w) a += red[w];
    red[16] = (float)(a / ((double)rows * cpg));
  }
  __syncthreads();
  const float mean = red[16];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it)
    if (live[it]) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) { const float d = (float)v[it][e] - mean; ss += d * d; }
    }
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) asm volatile("" : "+v"(v[it]));
  ss = wave_sum(ss);
  if (lane == 0) red[8 + wave] = ss;
  __syncthreads();
  if (tid == 0) {
    double b = 0.0;
    for (int w = 0; w < 8; ++w) b += red[8 + w];
    red[17] = (float)(1.0 / sqrt(b / ((double)rows * cpg) + (double)eps));
  }
  __syncthreads();
  const float rstd = red[17];
  float sc[VEC], sf[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const int ch = g * cpg + q * VEC + e;
    const float ga = gamma ? gamma[ch] : 1.f, be = beta ? beta[ch] : 0.f;
    sc[e] = rstd * ga;
    sf[e] = be - mean * sc[e];
  }
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int r = rr + it * rp;
    if (live[it]) {
      hv w;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float f = (float)v[it][e] * sc[e] + sf[e];
        if (silu) f = silu_f(f);
        w[e] = (f16)f;
      }
      *(hv *)(yb + (int64_t)r * c + q * VEC) = w;
    }
  }
}

// GroupNorm folded into the linear layer behind it (sp_groupnorm_fold_linear_f16): a block builds the instance's per-channel
// tables once in LDS (scale = gamma * rstd[group], mean[group], beta), then each of its waves walks output rows: the weight row
// times the scale, rounded to fp16, and the row's bias from the ROUNDED values (so that a constant offset of a group cancels
// exactly in the contraction).  instances * n * c halves written; no division and no global statistics read in the row loop.
constexpr int GN_FOLD_ROWS = 32;               // output rows per block (8 per wave)
__global__ __launch_bounds__(256) void gn_fold_linear_kernel(const float *__restrict__ stats, const f16 *__restrict__ w,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             const float *__restrict__ bias, f16 *__restrict__ w_out,
                                                             float *__restrict__ bias_out, int n, int c, int groups) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *t_scale = (float *)smem, *t_mean = t_scale + c, *t_beta = t_mean + c;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int inst = blockIdx.x;
  const int cpg = c / groups;
  const float *st = stats + (int64_t)inst * groups * 2;
  for (int ch = tid; ch < c; ch += 256) {
    const int g = ch / cpg;
    t_scale[ch] = (gamma ? gamma[ch] : 1.f) * st[g * 2 + 1];
    t_mean[ch] = st[g * 2];
    t_beta[ch] = beta ? beta[ch] : 0.f;
  }
  __syncthreads();
  const int oc = c >> 3;
  const int row0 = blockIdx.y * GN_FOLD_ROWS;
  for (int rr = wave; rr < GN_FOLD_ROWS; rr += 4) {
    const int row = row0 + rr;
    if (row >= n) break;
    const f16 *wr = w + (int64_t)row * c;
    f16 *wo = w_out + ((int64_t)inst * n + row) * c;
    float acc = 0.f;
    for (int o = lane; o < oc; o += 64) {
      const f16x8 v = *(const f16x8 *)(wr + o * 8);
      const f32x4 s0 = *(const f32x4 *)(t_scale + o * 8), s1 = *(const f32x4 *)(t_scale + o * 8 + 4);
      const f32x4 m0 = *(const f32x4 *)(t_mean + o * 8), m1 = *(const f32x4 *)(t_mean + o * 8 + 4);
      const f32x4 b0 = *(const f32x4 *)(t_beta + o * 8), b1 = *(const f32x4 *)(t_beta + o * 8 + 4);
      f16x8 q;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float w0 = (float)v[e], w1 = (float)v[e + 4];
        q[e] = (f16)(w0 * s0[e]);
        q[e + 4] = (f16)(w1 * s1[e]);
        acc += b0[e] * w0 - m0[e] * (float)q[e];
        acc += b1[e] * w1 - m1[e] * (float)q[e + 4];
      }
      *(f16x8 *)(wo + o * 8) = q;
    }
    acc = wave_sum(acc);
    if (lane == 0) bias_out[(int64_t)inst * n + row] = acc + (bias ? bias[row] : 0.f);
  }
}

// rows per statistics block: about 1024 blocks in all, >= 16 row-iterations per thread, at most 512 splits per instance
// (sp_groupnorm_ws_bytes), and a multiple of 4*P rows so that the kernel runs whole batches of four loads per thread
int64_t gn_rows_per_split(int instances, int64_t rows, int P) {
  int64_t want = (1024 + instances - 1) / instances;
  int64_t maxs = rows / (16 * (int64_t)P); if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want > 512) want = 512;
  if (want < 1) want = 1;
  const int64_t unit = 4 * (int64_t)P;
  int64_t per = (rows + want - 1) / want;
  per = (per + unit - 1) / unit * unit;
  return per;
}

// ---------------------------------------------------------------------------------- LayerNorm
// one wave per row; lane owns 8-channel octets lane, lane+64, ... (C/8 <= 64*NV)
template <int NV>
__global__ __launch_bounds__(256) void ln_kernel(const f16 *__restrict__ x, const f16 *__restrict__ addvec,
                                                 int64_t addvec_rows, f16 *__restrict__ sum_out,
                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                 f16 *__restrict__ y, int64_t rows, int c, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int oc = c >> 3;
  float v[NV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int o = lane + i * 64;
    if (o < oc) {
      f16x8 q = *(const f16x8 *)(x + row * c + o * 8);
      if (addvec) {
        const f16x8 a = *(const f16x8 *)(addvec + (row / addvec_rows) * c + o * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] = (f16)((float)q[e] + (float)a[e]);
        if (sum_out) *(f16x8 *)(sum_out + row * c + o * 8) = q;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = (float)q[e]; s += v[i][e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)c;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int o = lane + i * 64;
    if (o < oc) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; ss += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)c + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int o = lane + i * 64;
    if (o < oc) {
      const f32x4 g0 = *(const f32x4 *)(gamma + o * 8), g1 = *(const f32x4 *)(gamma + o * 8 + 4);
      const f32x4 b0 = *(const f32x4 *)(beta + o * 8), b1 = *(const f32x4 *)(beta + o * 8 + 4);
      f16x8 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        w[e] = (f16)((v[i][e] - mean) * rstd * g0[e] + b0[e]);
        w[e + 4] = (f16)((v[i][e + 4] - mean) * rstd * g1[e] + b1[e]);
      }
      *(f16x8 *)(y + row * c + o * 8) = w;
    }
  }
}

// statistics only (LayerNorm folded into the next GEMM), generic shape: one wave per row, lane owns 8-channel octets
// lane, lane+64, ... ; writes (mean, rstd)
template <int NV>
__global__ __launch_bounds__(256) void ln_stats_kernel(const f16 *__restrict__ x, const f16 *__restrict__ addvec,
                                                       int64_t addvec_rows, f16 *__restrict__ sum_out,
                                                       float *__restrict__ stats, int64_t rows, int c, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int oc = c >> 3;
  float v[NV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int o = lane + i * 64;
    if (o < oc) {
      f16x8 q = *(const f16x8 *)(x + row * c + o * 8);
      if (addvec) {
        const f16x8 a = *(const f16x8 *)(addvec + (row / addvec_rows) * c + o * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] = (f16)((float)q[e] + (float)a[e]);
        if (sum_out) *(f16x8 *)(sum_out + row * c + o * 8) = q;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = (float)q[e]; s += v[i][e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)c;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int o = lane + i * 64;
    if (o < oc) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; ss += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)c + eps);
  if (lane == 0) *(float2 *)(stats + row * 2) = make_float2(mean, rstd);
}

// The UNet's widths (C = 40 * LPR channels, LPR = 8 / 16 / 32): LPR lanes share a row, five 16-byte vectors each
// (vector j of a lane = octet j*LPR + lane_in_row: a wave instruction reads whole 128-byte lines), a wave covers
// 64 / LPR rows per pass and P passes with all 5 * P loads in flight; the row sums need log2(LPR) shuffle steps for
// 64 / LPR rows at once.  (One row per wave keeps 640 bytes in flight per wave and pays 12 shuffles per row:
// 2.4-2.9 TB/s at 129,024 rows.)
template <int LPR, int P>
__global__ __launch_bounds__(256) void ln_stats_rows_kernel(const f16 *__restrict__ x, const f16 *__restrict__ addvec,
                                                            int64_t addvec_rows, f16 *__restrict__ sum_out,
                                                            float *__restrict__ stats, int64_t rows, float eps) {
  constexpr int C = 40 * LPR, RPW = 64 / LPR;     // channels, rows per wave and pass
  const int lane = threadIdx.x & 63, lr = lane / LPR, li = lane % LPR;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (RPW * P);
  if (row0 >= rows) return;
  f16x8 q[P][5];
  int64_t row[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    row[p] = row0 + p * RPW + lr;
    const int64_t rr = row[p] < rows ? row[p] : rows - 1;         // (clamped rows are computed, not stored)
#pragma unroll
    for (int j = 0; j < 5; ++j) q[p][j] = *(const f16x8 *)(x + rr * C + (j * LPR + li) * 8);
  }
  if (addvec) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int64_t rr = row[p] < rows ? row[p] : rows - 1;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const f16x8 a = *(const f16x8 *)(addvec + (rr / addvec_rows) * C + (j * LPR + li) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) q[p][j][e] = (f16)((float)q[p][j][e] + (float)a[e]);
        if (sum_out && row[p] < rows) *(f16x8 *)(sum_out + rr * C + (j * LPR + li) * 8) = q[p][j];
      }
    }
  }
  float s[P], ss[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    s[p] = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) s[p] += (float)q[p][j][e];
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1)
#pragma unroll
    for (int p = 0; p < P; ++p) s[p] += __shfl_xor(s[p], o, 64);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    s[p] *= 1.0f / (float)C;
    ss[p] = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = (float)q[p][j][e] - s[p]; ss[p] += d * d; }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1)
#pragma unroll
    for (int p = 0; p < P; ++p) ss[p] += __shfl_xor(ss[p], o, 64);
#pragma unroll
  for (int p = 0; p < P; ++p)
    if (li == 0 && row[p] < rows) *(float2 *)(stats + row[p] * 2) = make_float2(s[p], rsqrtf(ss[p] * (1.0f / (float)C) + eps));
}

// (mean, rstd) per (instance, group) from the per-tile column sums a contraction's epilogue left (sp_gemm_desc.gn_part:
// [tile][half][c][2], 256-row tiles): one block per (instance, group) adds its tiles x halves x channels in a FIXED order
// (thread t takes every 256th entry, lane 0 of wave 0 folds the 256 partials) in fp64 -- sums of un-shifted values, so the
// cancellation of E[x^2] - mean^2 is left to the 53-bit fold (the per-tile sums are fp32 over 128 rows each).
// part_b != null: the normalised tensor is the CONCATENATION [a | b] of two producers' outputs (an up block's resnet reads
// [hidden | skip]): channels [0, c_a) come from part (row pitch c_a), channels [c_a, c) from part_b (row pitch c - c_a) --
// per-column sums are additive, so a group may straddle the seam.
__global__ __launch_bounds__(256) void gn_tile_sums_finalize_kernel(const float *__restrict__ part, float *__restrict__ stats,
                                                                   int tiles_per_inst, int c, int groups, int64_t rows,
                                                                   float eps, const float *__restrict__ part_b = nullptr,
                                                                   int c_a = 0) {
  __shared__ double red[256 * 2];
  const int inst = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  const int cpg = c / groups, ch0 = g * cpg;
  double a = 0.0, b = 0.0;
  // thread t takes the (tile, half) records t, t + 256, ...: the group's cpg (sum, sum of squares) pairs of a record are
  // contiguous (8 bytes each), every load of a thread is independent -- no index arithmetic beyond one multiply per record
  for (int rec = tid; rec < 2 * tiles_per_inst; rec += 256) {
    const int64_t r = (int64_t)inst * tiles_per_inst * 2 + rec;
    float sa = 0.f, sb = 0.f;
    if (!part_b) {
      const float2 *src = (const float2 *)(part + (r * c + ch0) * 2);
      for (int k = 0; k < cpg; ++k) { const float2 v = src[k]; sa += v.x; sb += v.y; }
    } else {
      for (int k = 0; k < cpg; ++k) {
        const int ch = ch0 + k;
        const float2 v = ch < c_a ? *(const float2 *)(part + (r * c_a + ch) * 2)
                                  : *(const float2 *)(part_b + (r * (c - c_a) + (ch - c_a)) * 2);
        sa += v.x; sb += v.y;
      }
    }
    a += (double)sa; b += (double)sb;          // (fp32 over one record's <= 128 channels-times-rows sums, fp64 across records)
  }
  red[tid * 2] = a; red[tid * 2 + 1] = b;
  __syncthreads();
  if (tid == 0) {
    a = 0.0; b = 0.0;
    const int n = 2 * tiles_per_inst < 256 ? 2 * tiles_per_inst : 256;
    for (int q = 0; q < n; ++q) { a += red[q * 2]; b += red[q * 2 + 1]; }
    const double cnt = (double)rows * cpg, mean = a / cnt;
    double var = b / cnt - mean * mean; if (var < 0.0) var = 0.0;
    stats[((int64_t)inst * groups + g) * 2] = (float)mean;
    stats[((int64_t)inst * groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

}  // namespace

extern "C" int sp_groupnorm_tile_sums2_f16(const void *x, int64_t ldx, const float *part, int c_a, const float *part_b,
                                           const float *gamma, const float *beta, void *y, int instances, int64_t rows, int c,
                                           int groups, float eps, int fuse_silu, float *stats, void *stream);

extern "C" int sp_groupnorm_tile_sums_f16(const void *x, int64_t ldx, const float *part, const float *gamma, const float *beta,
                                          void *y, int instances, int64_t rows, int c, int groups, float eps, int fuse_silu,
                                          float *stats, void *stream) {
  return sp_groupnorm_tile_sums2_f16(x, ldx, part, c, nullptr, gamma, beta, y, instances, rows, c, groups, eps, fuse_silu, stats,
                                     stream);
}

extern "C" int sp_groupnorm_tile_sums2_f16(const void *x, int64_t ldx, const float *part, int c_a, const float *part_b,
                                           const float *gamma, const float *beta, void *y, int instances, int64_t rows, int c,
                                           int groups, float eps, int fuse_silu, float *stats, void *stream) {
  SP_REQUIRE(x && y && part && stats, "sp_groupnorm_tile_sums_f16: null pointer");
  SP_REQUIRE(part_b ? (c_a > 0 && c_a < c && c_a % 4 == 0 && (c - c_a) % 4 == 0) : c_a == c,
             "sp_groupnorm_tile_sums_f16: c_a=%d invalid for C=%d", c_a, c);
  SP_REQUIRE(ldx >= c && ldx % 8 == 0, "sp_groupnorm_tile_sums_f16: ldx=%lld must be a multiple of 8 and >= C=%d", (long long)ldx, c);
  SP_REQUIRE(instances > 0 && rows > 0 && rows % 256 == 0 && rows / 256 <= 0x7fffffff,
             "sp_groupnorm_tile_sums_f16: an instance must be a whole number of 256-row tiles (rows=%lld)", (long long)rows);
  SP_REQUIRE(c % 8 == 0 && c >= 8 && c <= 4096, "sp_groupnorm_tile_sums_f16: C=%d must be a multiple of 8 in [8,4096]", c);
  SP_REQUIRE(groups > 0 && groups <= 64 && c % groups == 0, "sp_groupnorm_tile_sums_f16: groups=%d invalid for C=%d", groups, c);
  const int oc = c / 8;
  const int P = gn_rows_per_iter(oc);
  const int threads = ((oc * P + 63) / 64) * 64 < 64 ? 64 : ((oc * P + 63) / 64) * 64;
  SP_REQUIRE(threads <= 1024, "sp_groupnorm_tile_sums_f16: C too large");
  hipStream_t s = (hipStream_t)stream;
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(gn_tile_sums_finalize_kernel, dim3(instances, groups), dim3(256), 0, s, part, stats, (int)(rows / 256), c,
                     groups, rows, eps, part_b, c_a);
  SP_CHECK_LAUNCH("sp_groupnorm_tile_sums_f16(finalize)");
  int64_t blocks_y = (2048 + instances - 1) / instances;
  int64_t maxb = (rows + 16 * P - 1) / (16 * P);
  if (blocks_y > maxb) blocks_y = maxb;
  if (blocks_y < 1) blocks_y = 1;
  int64_t rpb = (rows + blocks_y - 1) / blocks_y;
  rpb = (rpb + 4 * P - 1) / (4 * P) * (4 * P);                 // whole batches of four loads per thread
  blocks_y = (rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(instances, (unsigned)blocks_y), dim3(threads), 0, s, (const f16 *)x,
                     (const float *)stats, (const float *)nullptr, 0, eps, gamma, beta, (f16 *)y, rows, c, groups, fuse_silu, rpb,
                     ldx);
  SP_CHECK_LAUNCH("sp_groupnorm_tile_sums_f16(apply)");
  return SP_OK;
}

// sp_groupnorm_fold_linear_f16 with the statistics taken from a producer's per-tile column sums instead of a pass over x
extern "C" int sp_groupnorm_fold_linear_tile_sums_f16(const float *part, const float *gamma, const float *beta, int instances,
                                                      int64_t rows, int c, int groups, float eps, const void *w,
                                                      const float *bias, int n, void *w_out, float *bias_out, float *stats,
                                                      void *stream) {
  SP_REQUIRE(part && w && w_out && bias_out && stats, "sp_groupnorm_fold_linear_tile_sums_f16: null pointer");
  SP_REQUIRE(instances > 0 && n > 0 && rows > 0 && rows % 256 == 0 && rows / 256 <= 0x7fffffff,
             "sp_groupnorm_fold_linear_tile_sums_f16: an instance must be a whole number of 256-row tiles (rows=%lld)", (long long)rows);
  SP_REQUIRE(c % 8 == 0 && c >= 8 && c <= 4096, "sp_groupnorm_fold_linear_tile_sums_f16: C=%d must be a multiple of 8 in [8,4096]", c);
  SP_REQUIRE(groups > 0 && groups <= 64 && c % groups == 0, "sp_groupnorm_fold_linear_tile_sums_f16: groups=%d invalid for C=%d", groups, c);
  hipStream_t s = (hipStream_t)stream;
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(gn_tile_sums_finalize_kernel, dim3(instances, groups), dim3(256), 0, s, part, stats, (int)(rows / 256), c,
                     groups, rows, eps);
  SP_CHECK_LAUNCH("sp_groupnorm_fold_linear_tile_sums_f16(finalize)");
  hipLaunchKernelGGL(gn_fold_linear_kernel, dim3(instances, (n + GN_FOLD_ROWS - 1) / GN_FOLD_ROWS), dim3(256),
                     (size_t)3 * c * sizeof(float), s, (const float *)stats, (const f16 *)w, gamma, beta, bias, (f16 *)w_out,
                     bias_out, n, c, groups);
  SP_CHECK_LAUNCH("sp_groupnorm_fold_linear_tile_sums_f16(fold)");
  return SP_OK;
}

extern "C" size_t sp_groupnorm_ws_bytes(int instances, int64_t rows, int c, int groups) {
  if (instances <= 0 || rows <= 0 || c <= 0 || groups <= 0) return 0;
  return (size_t)instances * (512 + 1) * groups * 2 * sizeof(float);
}

extern "C" int sp_groupnorm_ld_f16(const void *x, int64_t ldx, const float *gamma, const float *beta, void *y,
                                   int instances, int64_t rows, int c, int groups, float eps,
                                   int fuse_silu, void *ws, size_t ws_bytes, void *stream);

extern "C" int sp_groupnorm_f16(const void *x, const float *gamma, const float *beta, void *y,
                                int instances, int64_t rows, int c, int groups, float eps,
                                int fuse_silu, void *ws, size_t ws_bytes, void *stream) {
  return sp_groupnorm_ld_f16(x, c, gamma, beta, y, instances, rows, c, groups, eps, fuse_silu, ws, ws_bytes, stream);
}

// x: [instances * rows][ldx] with the C normalised channels at the head of every row (a column slice of a wider
// tensor: the skip half of a concatenation buffer); y stays dense [instances * rows][C]
extern "C" int sp_groupnorm_ld_f16(const void *x, int64_t ldx, const float *gamma, const float *beta, void *y,
                                   int instances, int64_t rows, int c, int groups, float eps,
                                   int fuse_silu, void *ws, size_t ws_bytes, void *stream) {
  SP_REQUIRE(x && y && ws, "sp_groupnorm_f16: null pointer");
  SP_REQUIRE(ldx >= c && ldx % 8 == 0, "sp_groupnorm_f16: ldx=%lld must be a multiple of 8 and >= C=%d", (long long)ldx, c);
  SP_REQUIRE(instances > 0 && rows > 0, "sp_groupnorm_f16: instances/rows must be positive");
  SP_REQUIRE(c % 8 == 0 && c >= 8 && c <= 4096, "sp_groupnorm_f16: C=%d must be a multiple of 8 in [8,4096]", c);
  SP_REQUIRE(groups > 0 && groups <= 64 && c % groups == 0, "sp_groupnorm_f16: groups=%d invalid for C=%d", groups, c);
  SP_REQUIRE(ws_bytes >= sp_groupnorm_ws_bytes(instances, rows, c, groups), "sp_groupnorm_f16: workspace too small");
  {
    // slab of one (instance, group) small enough for 512 threads x 128 halves of registers -> single fused launch
    const int cpg = c / groups;
    hipStream_t fs = (hipStream_t)stream;
    // (an 8-byte-vector variant for 20/60 channels per group measured slower than the three-launch path)
    // At least 128 workgroups -- or a tensor so small (<= 8 MB: the 2,016-row level normalised over all frames, 32
    // workgroups) that three launches cost more than the 32 CUs' streaming time.
    if (cpg % 8 == 0 && cpg / 8 <= 512 && (int64_t)instances * groups <= 0x7fffffff &&
        ((int64_t)instances * groups >= 128 || (int64_t)instances * rows * c * 2 <= (8 << 20))) {
      const int64_t rp = 512 / (cpg / 8);
      SP_CLEAR_STALE_ERROR();
      if (rows <= 16 * rp) {
        hipLaunchKernelGGL((gn_fused_kernel<8, 16>), dim3(instances * groups), dim3(512), 0, fs, (const f16 *)x, gamma,
                           beta, (f16 *)y, rows, c, groups, eps, fuse_silu, ldx);
        SP_CHECK_LAUNCH("sp_groupnorm_f16(fused)");
        return SP_OK;
      }
      if (rows <= 20 * rp) {
        hipLaunchKernelGGL((gn_fused_kernel<8, 20>), dim3(instances * groups), dim3(512), 0, fs, (const f16 *)x, gamma,
                           beta, (f16 *)y, rows, c, groups, eps, fuse_silu, ldx);
        SP_CHECK_LAUNCH("sp_groupnorm_f16(fused)");
        return SP_OK;
      }
    }
  }
  const int oc = c / 8;
  const int P = gn_rows_per_iter(oc);
  const int threads = ((oc * P + 63) / 64) * 64 < 64 ? 64 : ((oc * P + 63) / 64) * 64;
  SP_REQUIRE(threads <= 1024, "sp_groupnorm_f16: C too large");
  const int64_t per = gn_rows_per_split(instances, rows, P);
  const int splits = (int)((rows + per - 1) / per);            // <= 512
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = ((size_t)P * c * 2 + (size_t)(threads / groups) * groups * 2) * sizeof(float);
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(gn_stats_kernel, dim3(instances, splits), dim3(threads), lds, s, (const f16 *)x,
                     (float *)ws, rows, c, groups, splits, per, ldx);
  SP_CHECK_LAUNCH("sp_groupnorm_f16(stats)");
  int64_t blocks_y = (2048 + instances - 1) / instances;
  int64_t maxb = (rows + 16 * P - 1) / (16 * P);
  if (blocks_y > maxb) blocks_y = maxb;
  if (blocks_y < 1) blocks_y = 1;
  int64_t rpb = (rows + blocks_y - 1) / blocks_y;
  rpb = (rpb + 4 * P - 1) / (4 * P) * (4 * P);                 // whole batches of four loads per thread
  blocks_y = (rows + rpb - 1) / rpb;
  SP_CLEAR_STALE_ERROR();
  float *stats = (float *)ws + (size_t)instances * splits * groups * 2;
  // Few splits per instance (the per-frame norms: 74 at 14 frames): every apply block folds its instance's partials in
  // its prologue (19 KB of L2 reads per block) and the finalize launch between the two passes is gone.  Many splits
  // (one instance over all frames: 512 x 32 partials = 131 KB per apply block) keep the one-block-per-instance fold.
  const bool fold_in_apply = splits <= 128;
  if (!fold_in_apply)
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(instances), dim3(1024), 0, s, (const f16 *)x, (const float *)ws, stats,
                       rows, c, groups, splits, eps, ldx);
  const size_t lds_apply = fold_in_apply ? (size_t)(threads / groups) * groups * 2 * sizeof(double) + groups * 2 * sizeof(float) : 0;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(instances, (unsigned)blocks_y), dim3(threads), lds_apply, s,
                     (const f16 *)x, (const float *)stats, fold_in_apply ? (const float *)ws : (const float *)nullptr, splits,
                     eps, gamma, beta, (f16 *)y, rows, c, groups, fuse_silu, rpb, ldx);
  SP_CHECK_LAUNCH("sp_groupnorm_f16(apply)");
  return SP_OK;
}

// GroupNorm (no activation) folded into the linear layer behind it: statistics pass + one small kernel (svdpipe.h)
extern "C" int sp_groupnorm_fold_linear_f16(const void *x, int64_t ldx, const float *gamma, const float *beta, int instances,
                                            int64_t rows, int c, int groups, float eps, const void *w, const float *bias,
                                            int n, void *w_out, float *bias_out, void *ws, size_t ws_bytes, void *stream) {
  SP_REQUIRE(x && w && w_out && bias_out && ws, "sp_groupnorm_fold_linear_f16: null pointer");
  SP_REQUIRE(ldx >= c && ldx % 8 == 0, "sp_groupnorm_fold_linear_f16: ldx=%lld must be a multiple of 8 and >= C=%d",
             (long long)ldx, c);
  SP_REQUIRE(instances > 0 && rows > 0 && n > 0, "sp_groupnorm_fold_linear_f16: instances/rows/n must be positive");
  SP_REQUIRE(c % 8 == 0 && c >= 8 && c <= 4096, "sp_groupnorm_fold_linear_f16: C=%d must be a multiple of 8 in [8,4096]", c);
  SP_REQUIRE(groups > 0 && groups <= 64 && c % groups == 0, "sp_groupnorm_fold_linear_f16: groups=%d invalid for C=%d", groups, c);
  SP_REQUIRE(ws_bytes >= sp_groupnorm_ws_bytes(instances, rows, c, groups), "sp_groupnorm_fold_linear_f16: workspace too small");
  const int oc = c / 8;
  const int P = gn_rows_per_iter(oc);
  const int threads = ((oc * P + 63) / 64) * 64 < 64 ? 64 : ((oc * P + 63) / 64) * 64;
  SP_REQUIRE(threads <= 1024, "sp_groupnorm_fold_linear_f16: C too large");
  const int64_t per = gn_rows_per_split(instances, rows, P);
  const int splits = (int)((rows + per - 1) / per);            // <= 512
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = ((size_t)P * c * 2 + (size_t)(threads / groups) * groups * 2) * sizeof(float);
  float *stats = (float *)ws + (size_t)instances * splits * groups * 2;
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(gn_stats_kernel, dim3(instances, splits), dim3(threads), lds, s, (const f16 *)x, (float *)ws, rows, c,
                     groups, splits, per, ldx);
  SP_CHECK_LAUNCH("sp_groupnorm_fold_linear_f16(stats)");
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(instances), dim3(1024), 0, s, (const f16 *)x, (const float *)ws, stats, rows, c,
                     groups, splits, eps, ldx);
  SP_CHECK_LAUNCH("sp_groupnorm_fold_linear_f16(finalize)");
  hipLaunchKernelGGL(gn_fold_linear_kernel, dim3(instances, (n + GN_FOLD_ROWS - 1) / GN_FOLD_ROWS), dim3(256),
                     (size_t)3 * c * sizeof(float), s, (const float *)stats, (const f16 *)w, gamma, beta, bias, (f16 *)w_out,
                     bias_out, n, c, groups);
  SP_CHECK_LAUNCH("sp_groupnorm_fold_linear_f16(fold)");
  return SP_OK;
}

extern "C" int sp_layernorm_f16(const void *x, const void *addvec, int64_t addvec_rows, void *sum_out,
                                const float *gamma, const float *beta, void *y, int64_t rows, int c,
                                float eps, void *stream) {
  SP_REQUIRE(x && y && gamma && beta, "sp_layernorm_f16: null pointer");
  SP_REQUIRE(rows > 0 && c % 8 == 0 && c >= 8 && c <= 2048, "sp_layernorm_f16: rows=%lld C=%d unsupported",
             (long long)rows, c);
  if (addvec) SP_REQUIRE(addvec_rows > 0, "sp_layernorm_f16: addvec_rows must be positive");
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  const int oc = c / 8;
  SP_CLEAR_STALE_ERROR();
#define LN_LAUNCH(NV)                                                                              \
  hipLaunchKernelGGL(ln_kernel<NV>, dim3(grid), dim3(256), 0, s, (const f16 *)x, (const f16 *)addvec, \
                     addvec_rows > 0 ? addvec_rows : 1, (f16 *)sum_out, gamma, beta, (f16 *)y, rows, c, eps)
  if (oc <= 64) LN_LAUNCH(1);
  else if (oc <= 128) LN_LAUNCH(2);
  else if (oc <= 192) LN_LAUNCH(3);
  else LN_LAUNCH(4);
#undef LN_LAUNCH
  SP_CHECK_LAUNCH("sp_layernorm_f16");
  return SP_OK;
}

extern "C" int sp_ln_stats_f16(const void *x, const void *addvec, int64_t addvec_rows, void *sum_out, float *stats,
                               int64_t rows, int c, float eps, void *stream) {
  SP_REQUIRE(x && stats, "sp_ln_stats_f16: null pointer");
  SP_REQUIRE(rows > 0 && c % 8 == 0 && c >= 8 && c <= 2048, "sp_ln_stats_f16: rows=%lld C=%d unsupported",
             (long long)rows, c);
  if (addvec) SP_REQUIRE(addvec_rows > 0, "sp_ln_stats_f16: addvec_rows must be positive");
  hipStream_t s = (hipStream_t)stream;
  const int oc = c / 8;
  SP_CLEAR_STALE_ERROR();
#define LNS_ROWS(LPR, P)                                                                                       \
  hipLaunchKernelGGL((ln_stats_rows_kernel<LPR, P>),                                                           \
                     dim3((unsigned)((rows + 4 * (64 / LPR) * P - 1) / (4 * (64 / LPR) * P))), dim3(256), 0, s, \
                     (const f16 *)x, (const f16 *)addvec, addvec_rows > 0 ? addvec_rows : 1, (f16 *)sum_out,   \
                     stats, rows, eps)
#define LNS_LAUNCH(NV)                                                                                    \
  hipLaunchKernelGGL(ln_stats_kernel<NV>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const f16 *)x, \
                     (const f16 *)addvec, addvec_rows > 0 ? addvec_rows : 1, (f16 *)sum_out, stats, rows, c, eps)
  if (c == 320 && rows >= 8192) LNS_ROWS(8, 2);
  else if (c == 640 && rows >= 4096) LNS_ROWS(16, 2);
  else if (c == 1280 && rows >= 2048) LNS_ROWS(32, 2);
  else if (oc <= 64) LNS_LAUNCH(1);
  else if (oc <= 128) LNS_LAUNCH(2);
  else if (oc <= 192) LNS_LAUNCH(3);
  else LNS_LAUNCH(4);
#undef LNS_ROWS
#undef LNS_LAUNCH
  SP_CHECK_LAUNCH("sp_ln_stats_f16");
  return SP_OK;
}

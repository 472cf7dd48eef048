// Kernels only the CLIP image encoder on the first stage needs (SURVEY.md 8f-3;
// /root/reference/scripts/generate_video_demo.py:108-112: image_encoder(pixel_values).image_embeds with
// transformers.CLIPVisionModelWithProjection, ViT-H/14: 257 tokens, width 1280, 16 heads of 80, 32 layers).  Its
// projections, MLP contractions and LayerNorms are the UNet's kernels (gemm*.hip, norm.hip, elementwise.hip); a
// forward is 0.33 TFLOP once per video, so these three are written for simplicity, not for the roofline.
//
//   patchify_kernel   : (B,3,H,W) pixels -> im2col rows [B*(H/P)*(W/P)][kpad] (k = c*P*P + ky*P + kx: the flattening
//                       of the patch-embedding Conv2d weight), zero padded to the GEMM's K granule.
//   attn_small_kernel : softmax(q k^T * scale) v for SHORT sequences and any head width (ViT: 257 tokens, 80 wide):
//                       K and V of one (image, head) live in LDS, a wave owns a query row at a time; fp32 math.
//   gelu_kernel       : exact (erf) or "quick" GELU between fc1 and fc2.
#include "common.h"

namespace {

__global__ void patchify_kernel(const f16 *__restrict__ px, f16 *__restrict__ rows, int h, int w, int patch, int kpad,
                                int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;    // one output element
  if (idx >= total) return;
  const int k = (int)(idx % kpad);
  const int64_t r = idx / kpad;
  const int pw = w / patch, ph = h / patch;
  const int pxi = (int)(r % pw), pyi = (int)((r / pw) % ph);
  const int64_t b = r / ((int64_t)pw * ph);
  f16 v = (f16)0.f;
  if (k < 3 * patch * patch) {
    const int c = k / (patch * patch), rem = k - c * patch * patch;
    const int ky = rem / patch, kx = rem - ky * patch;
    v = px[((b * 3 + c) * h + pyi * patch + ky) * (int64_t)w + pxi * patch + kx];
  }
  rows[idx] = v;
}

constexpr int AS_ROWS = 32;      // query rows per workgroup (8 per wave)
constexpr int AS_MAXK = 8;       // keys per lane: seq <= 512

__global__ __launch_bounds__(256) void attn_small_kernel(const f16 *__restrict__ q, const f16 *__restrict__ k,
                                                         const f16 *__restrict__ v, f16 *__restrict__ o, int64_t ldq,
                                                         int64_t ldk, int64_t ldv, int64_t ldo, int seq, int heads, int hd,
                                                         float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int hd2 = hd >> 1, ks = hd2 | 1;                 // K row pitch in dwords, odd: lanes on different rows hit
  unsigned *kl = (unsigned *)smem;                       // different banks
  unsigned *vl = kl + (size_t)seq * ks;                  // V: [seq][hd2] dwords
  float *pl = (float *)(vl + (size_t)seq * hd2);         // per wave: probabilities [seq]
  float *ql = pl + 4 * (size_t)seq;                      // per wave: scaled query [hd]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / heads, hh = bh - b * heads;
  const int64_t row0 = (int64_t)b * seq;
  // ---- K, V of this (image, head) -> LDS
  const int chunks = hd >> 3;                            // 16-byte pieces per row
  for (int i = tid; i < seq * chunks; i += 256) {
    const int r = i / chunks, c = i - r * chunks;
    const uint4 kv = *(const uint4 *)(k + (row0 + r) * ldk + hh * hd + c * 8);
    const uint4 vv = *(const uint4 *)(v + (row0 + r) * ldv + hh * hd + c * 8);
    unsigned *kd = kl + (size_t)r * ks + c * 4, *vd = vl + (size_t)r * hd2 + c * 4;
    kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
    vd[0] = vv.x; vd[1] = vv.y; vd[2] = vv.z; vd[3] = vv.w;
  }
  __syncthreads();
  float *pw = pl + (size_t)wave * seq, *qw = ql + (size_t)wave * hd;
  const int r_end = min((int)(blockIdx.x + 1) * AS_ROWS, seq);
  for (int i = blockIdx.x * AS_ROWS + wave; i < r_end; i += 4) {
    for (int d = lane; d < hd; d += 64) qw[d] = (float)q[(row0 + i) * ldq + hh * hd + d] * scale;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // scores of the keys lane, lane+64, ...
    float s[AS_MAXK];
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < AS_MAXK; ++j) {
      const int kk = lane + j * 64;
      s[j] = -3.0e38f;
      if (kk < seq) {
        const unsigned *kr = kl + (size_t)kk * ks;
        float acc = 0.f;
        for (int d2 = 0; d2 < hd2; ++d2) {
          const f16x2 kk2 = __builtin_bit_cast(f16x2, kr[d2]);
          acc = fmaf((float)kk2[0], qw[2 * d2], acc);
          acc = fmaf((float)kk2[1], qw[2 * d2 + 1], acc);
        }
        s[j] = acc;
        mx = fmaxf(mx, acc);
      }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < AS_MAXK; ++j) {
      const int kk = lane + j * 64;
      if (kk < seq) {
        const float p = __builtin_amdgcn_exp2f((s[j] - mx) * 1.4426950408889634f);
        sum += p;
        pw[kk] = p;
      }
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // output channels 2*lane, 2*lane+1 (and + 128 when hd > 128: not supported, hd <= 128)
    if (lane < hd2) {
      float a0 = 0.f, a1 = 0.f;
      for (int kk = 0; kk < seq; ++kk) {
        const f16x2 vv = __builtin_bit_cast(f16x2, vl[(size_t)kk * hd2 + lane]);
        const float p = pw[kk];
        a0 = fmaf(p, (float)vv[0], a0);
        a1 = fmaf(p, (float)vv[1], a1);
      }
      const float inv = 1.0f / sum;
      const f16x2 w = {(f16)(a0 * inv), (f16)(a1 * inv)};
      *(f16x2 *)(o + (row0 + i) * ldo + hh * hd + 2 * lane) = w;
    }
    __builtin_amdgcn_wave_barrier();                      // pw / qw are rewritten by the next row
  }
}

__global__ void gelu_kernel(const f16 *__restrict__ x, f16 *__restrict__ y, int64_t n8, int quick) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n8) return;
  const f16x8 v = *(const f16x8 *)(x + idx * 8);
  f16x8 w;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float f = (float)v[e];
    w[e] = (f16)(quick ? f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * f)) : gelu_f(f));
  }
  *(f16x8 *)(y + idx * 8) = w;
}

}  // namespace

extern "C" int sp_patchify_f16(const void *pixels, void *rows, int batch, int h, int w, int patch, int kpad, void *stream) {
  SP_REQUIRE(pixels && rows, "sp_patchify_f16: null pointer");
  SP_REQUIRE(batch > 0 && patch > 0 && h > 0 && w > 0 && h % patch == 0 && w % patch == 0,
             "sp_patchify_f16: image %dx%d must be whole patches of %d", h, w, patch);
  SP_REQUIRE(kpad >= 3 * patch * patch, "sp_patchify_f16: kpad=%d smaller than 3*patch*patch", kpad);
  const int64_t total = (int64_t)batch * (h / patch) * (w / patch) * kpad;
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const f16 *)pixels, (f16 *)rows, h, w, patch, kpad, total);
  SP_CHECK_LAUNCH("sp_patchify_f16");
  return SP_OK;
}

extern "C" int sp_attn_small_f16(const void *q, const void *k, const void *v, void *o, int64_t ldq, int64_t ldk,
                                 int64_t ldv, int64_t ldo, int batch, int seq, int heads, int head_dim, float scale,
                                 void *stream) {
  SP_REQUIRE(q && k && v && o, "sp_attn_small_f16: null pointer");
  SP_REQUIRE(batch > 0 && heads > 0 && seq > 0 && seq <= 64 * AS_MAXK, "sp_attn_small_f16: seq=%d must be in [1,%d]", seq,
             64 * AS_MAXK);
  SP_REQUIRE(head_dim >= 8 && head_dim <= 128 && head_dim % 8 == 0, "sp_attn_small_f16: head_dim=%d must be a multiple of 8 in [8,128]", head_dim);
  SP_REQUIRE(ldq % 2 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 2 == 0, "sp_attn_small_f16: strides (ldk, ldv multiples of 8)");
  SP_REQUIRE((int64_t)batch * heads <= 65535, "sp_attn_small_f16: batch*heads too large");
  const int hd2 = head_dim / 2, ks = hd2 | 1;
  const size_t lds = ((size_t)seq * ks + (size_t)seq * hd2 + 4 * (size_t)seq + 4 * (size_t)head_dim) * 4;
  SP_REQUIRE(lds <= 160 * 1024, "sp_attn_small_f16: K and V of one head (%zu bytes) do not fit in LDS", lds);
  static bool attr_set[SP_MAX_DEVICES] = {};
  if (int rc = sp_ensure_dyn_lds((const void *)attn_small_kernel, (int)(160 * 1024), attr_set, "sp_attn_small_f16")) return rc;
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(attn_small_kernel, dim3((seq + AS_ROWS - 1) / AS_ROWS, batch * heads), dim3(256), lds,
                     (hipStream_t)stream, (const f16 *)q, (const f16 *)k, (const f16 *)v, (f16 *)o, ldq, ldk, ldv, ldo, seq,
                     heads, head_dim, scale);
  SP_CHECK_LAUNCH("sp_attn_small_f16");
  return SP_OK;
}

extern "C" int sp_gelu_f16(const void *x, void *y, int64_t n, int quick, void *stream) {
  SP_REQUIRE(x && y && n > 0 && n % 8 == 0, "sp_gelu_f16: n=%lld must be a positive multiple of 8", (long long)n);
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f16 *)x,
                     (f16 *)y, n / 8, quick);
  SP_CHECK_LAUNCH("sp_gelu_f16");
  return SP_OK;
}

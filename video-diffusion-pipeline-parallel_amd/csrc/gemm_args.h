// Shared by the implicit-GEMM kernels (gemm.hip, gemm_pp.hip).
#pragma once
#include "common.h"

namespace spgemm {

constexpr int BK = 64;

template <int N>
__device__ __forceinline__ void wait_vm_lgkm() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

struct GemmArgs {
  const f16 *a;
  const f16 *a2;            // extra LINEAR tap behind the regular ones (sp_gemm_desc.a2): rows [m][lda2], cin2 channels
  int64_t lda2;
  int cin2;
  const f16 *w;
  const float *bias;
  const float *bias2;
  const f16 *res1;
  const f16 *res2;
  // Euler tail (conv_out only): latent in / out (B,4,F,H,W) fp16, optional unconditional eps rows + per-frame guidance
  const f16 *eul_lat; f16 *eul_out; const f16 *eul_u; const float *eul_gs;
  int64_t eul_ldu, eul_hw;
  int eul_frames;
  float eul_c_out, eul_c_skip, eul_inv_sigma, eul_dt;
  const float *ln_stats;    // LayerNorm fold: fp32 [m][2] (mean, rstd), or null
  const float *ln_colsum;   // fp32 [n]: row sums of the gamma-scaled weight
  float *ln_out;            // fp32 [m][2]: (mean, rstd) of the stored output rows, or null (ping-pong kernels, n == BN)
  float ln_out_eps;
  float *gn_part;           // fp32 [tiles_m][2][n][2]: per 128-row half of every 256-row tile, per output column, (sum, sum of
                            // squares) of the fp32 outputs (GroupNorm statistics of the NEXT norm: sp_gemm_desc.gn_part)
  float *ln_part;           // rows spanning two tiles: fp32 [m][tiles_n][2] raw (sum, sum of squares) per tile, folded by ln_part_finalize
  f16 *d;
  const char *zero;
  int64_t lda, ldr1, ldr2, ldd, hw, bias2_rows, ldb2;
  int64_t w_group_rows, w_group_stride;   // > 0: rows [g*w_group_rows, ..) use the weights at w + g*w_group_stride (ping-pong kernels)
  int mode, cin, taps;
  int n_img, hin, win, hout, wout, stride, ups;
  int frames;
  int m, n, k;
  float oscale, r1scale, r2scale;
  int geglu, n_store;
  int tiles_m, tiles_n;
  int ksplit;          // > 1: K is cut into ksplit slices, fp32 partial sums go to `partial` ([ksplit][m][n])
  float *partial;
  int dbg;   // ablation builds only (-DSP_GEMM_EXPERIMENTS + SP_GEMM_DBG): selects gemm_pp_kernel<.., EXP>
  int stagger;   // ablation builds only (SP_GEMM_STAGGER): first-round workgroups start (b/8 & 3) * stagger us late
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // bijective "each XCD gets a contiguous run" remap (blocks b and b+8 share an XCD)
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}


// name of the kernel instantiation the calling thread launched last (sp_gemm_last_kernel: profile summaries)
void note_kernel(const char *fmt, ...);
void note_kernel_suffix(const char *suffix);   // appended to the name note_kernel left (a follow-up launch of the same call)

// ping-pong large-tile kernels (gemm_pp.hip): bm in {128 (bn 256 only), 192, 256}, bn in {256, 320}
int launch_pp(GemmArgs &a, int bm, int bn, hipStream_t s);

// persistent-stream kernels (gemm_ps.hip): (bm, bn) in {(256,256), (192,256)}; see ps_supported
bool ps_supported(const GemmArgs &a, int bm, int bn);
int launch_ps(GemmArgs &a, int bm, int bn, hipStream_t s);

}  // namespace spgemm

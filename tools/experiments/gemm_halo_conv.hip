// EXPERIMENT, not built into the library: halo-tile 3x3 convolution for the ping-pong GEMM family.
//
// Measured on MI355X (tools/bench_routes.py with a forced route 5, gpurun_out -> profiles/r03j_halo_conv.txt), against
// gemm_pp_kernel<256, 320 / 256> on the UNet's own stride-1 convolutions at two videos per call:
//   258,048 rows (72 x 128): 320->320  418 vs 397 us (1,137 vs 1,197 TFLOP/s), 640->320  782 vs 763, 960->320  1,145 vs 1,120
//    64,512 rows (36 x 64):  640->640  394 vs 366 us (1,208 vs 1,300),         1,280->640  738 vs 698, 1,280->... 831 vs 742
// i.e. 3-10 % SLOWER, although it issues 3.6x fewer activation LDS-DMA pieces (40 instead of 144 per 32 channels) and
// although a timing-only build of gemm_pp_kernel that merely skipped the activation pieces of taps 1-8 had promised
// x1.07-1.24: what the real kernel adds -- five vector instructions per activation fragment and K-step for the shifted,
// re-swizzled window address (the ping-pong kernel's fragment addresses are loop constants), and the K order chunk-major
// / tap-minor, which walks the weights with a stride of Cin instead of streaming them -- costs more than the pieces
// saved.  These convolutions already run at 1.2-1.37 PFLOP/s in isolation; the contraction family's weak shapes are the
// K = 320 linears, not the convolutions.  Results were correct (seven shapes against fp32 torch
// and against the ping-pong kernel, rel-L2 <= 1e-3: the test is quoted at the end).
//
// How it was plugged in: the kernel below sat in csrc/gemm_pp.hip in front of launch_pp<> (it uses that file's
// pp_epilogue, wait_dma, wait_vm, swz4, PBK), halo_supported / launch_halo were declared in gemm_args.h, and
// gemm.hip::dispatch started with `if (route == 5 && halo_supported(a)) return launch_halo(a, s);`.

// ------------------------------------------------------------------------------------------------------------------
// Halo-tile 3x3 convolution (stride 1, no upsample, image width W in {64, 128}, rows-per-image a multiple of 256):
// gemm_pp_kernel fetches the activation rows of a tile once per TAP (nine 16-piece LDS-DMA rounds per 32 channels);
// here a tile = 256 / W whole image rows, and the (256 / W + 2) x (W + 2) input pixels it touches are brought into LDS
// ONCE per 32 channels (40 / 32 one-KiB pieces instead of 144), the nine taps read shifted windows of that image.
//   K order: 32-channel chunk major, tap minor; weights stream as before (one BN x 64 B stage per K-step, 4-deep ring).
//   Halo image: pixel hp = hy * (W + 2) + hx at byte hp * 64, 16-byte chunk c at chunk c ^ ((hp >> 2) & 3) (applied on
//   the DMA source address): a 16-row operand fragment = 16 consecutive pixels from ANY start, still conflict-free.
//   Two halo buffers: chunk c+1 arrives (one piece per wave per K-step, iterations t < HL) while chunk c is read.
//   vmcnt: a K-step's pieces are issued [halo piece, weight pieces]; "K-step k+1 landed" = at most the pieces issued
//   after its weight pieces outstanding = the weight pieces of the later K-steps + the halo pieces issued beside them.
template <int BN, int W>
__global__ __launch_bounds__(512, 2) void gemm_halo_kernel(const GemmArgs p) {
  constexpr int BM = 256, NWV = 8, NT = 512, PSTAGES = 4, PDIST = 3;
  constexpr int TN = BN / 4 / 16, TM = BM / 2 / 16, WTN = BN / 4, WTM = BM / 2;
  constexpr int B_BYTES = BN * 64, B_PIECES = BN / 16;
  constexpr int B_LOADS_LO = (B_PIECES + NWV - 1) / NWV, B_LOADS_HI = B_PIECES / NWV;
  constexpr int B_SPLIT = B_PIECES % NWV == 0 ? NWV : B_PIECES % NWV;
  static_assert(B_SPLIT == NWV || B_SPLIT == 4, "wave halves must have uniform DMA counts");
  constexpr int L_EARLY = B_LOADS_LO, L_LATE = B_SPLIT == NWV ? B_LOADS_LO : B_LOADS_HI;
  constexpr int HW = W + 2, R = BM / W, HPIX = (R + 2) * HW;
  constexpr int HL = (HPIX + 16 * NWV - 1) / (16 * NWV);        // halo pieces per wave per chunk
  constexpr int HALO_BYTES = HL * NWV * 1024, RING = PSTAGES * B_BYTES;
  static_assert(HL <= 5, "the halo of chunk c+1 is issued in iterations 0..HL-1 of chunk c, ahead of K-step (c+1, 0)'s weights");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const bool late = wave >= 4;

  constexpr int GM = 4;
  const int nwg = p.tiles_m * p.tiles_n;
  const int t = xcd_remap(blockIdx.x, nwg);
  const int per_group = GM * p.tiles_n;
  const int group = t / per_group;
  const int first_m = group * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int in_group = t - group * per_group;
  const int tile_n = in_group / gsz;
  const int tile_m = first_m + (in_group - tile_n * gsz);

  // ---------------------------------------------------------------- producer state
  const int lrow = lane >> 2, lchunk = lane & 3;
  const int cpt = p.cin >> 5;                     // 32-channel chunks
  const int nk = 9 * cpt;
  const int per_img = p.hin * W;                  // (host: win == W, a multiple of 256 rows per image)
  const int img = (tile_m * BM) / per_img;
  const int ty0 = ((tile_m * BM) - img * per_img) / W;
  const f16 *hptr[HL];                            // this lane's 16 bytes of halo piece i (chunk 0); +32 halves per chunk
  unsigned hvalid = 0;                            // bit i: piece i of this lane is inside the image (else: zero page, no step)
#pragma unroll
  for (int i = 0; i < HL; ++i) {
    const int hp = (i * NWV + wave) * 16 + lrow;
    const int hy = hp / HW, hx = hp - hy * HW;
    const int y = ty0 - 1 + hy, x = hx - 1;
    if (hp < HPIX && y >= 0 && y < p.hin && x >= 0 && x < W) {
      hptr[i] = p.a + ((int64_t)(img * p.hin + y) * W + x) * p.lda + ((lchunk ^ ((hp >> 2) & 3)) << 3);
      hvalid |= 1u << i;
    } else {
      hptr[i] = (const f16 *)(p.zero + lchunk * 16);
    }
  }
  const f16 *bbase[B_LOADS_LO];
#pragma unroll
  for (int j = 0; j < B_LOADS_LO; ++j) {
    const int r = (j * NWV + wave) * 16 + lrow;
    const int n = tile_n * BN + (r < BN ? r : 0);
    bbase[j] = p.w + (int64_t)n * p.k + (lchunk ^ swz4(r)) * 8;
  }
  // weights of K-step (chunk sc, tap st) into ring slot
  int sc = 0, st = 0, stage_slot = 0;
  auto stage_w = [&]() {
    char *sb = smem + stage_slot * B_BYTES;
    const int koff = st * p.cin + (sc << 5);
#pragma unroll
    for (int j = 0; j < B_LOADS_LO; ++j)
      if (j < B_LOADS_HI || wave < B_SPLIT) glds16(bbase[j] + koff, sb + (j * NWV + wave) * 1024);
    stage_slot = stage_slot + 1 == PSTAGES ? 0 : stage_slot + 1;
    if (++st == 9) { st = 0; ++sc; }
  };
  // piece i of the halo of the chunk after the one hptr points at
  auto stage_h = [&](int i, int buf) {
    char *dst = smem + RING + buf * HALO_BYTES;
#pragma unroll
    for (int q = 0; q < HL; ++q)
      if (q == i) { glds16(hptr[q], dst + (q * NWV + wave) * 1024); hptr[q] += (hvalid >> q) & 1 ? PBK : 0; }
  };

  // ---------------------------------------------------------------- accumulators start at bias (+ time-embedding row)
  f32x4 acc[TN][TM], bias_v[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    bias_v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias_v[i] = *(const f32x4 *)(p.bias + tile_n * BN + wn * WTN + i * 16 + 4 * (lane >> 4));
  }
  __builtin_amdgcn_sched_barrier(0);

  const int fr = lane & 15, fq = lane >> 4;
  // fragment i of the weights: + i * 1024 bytes; fragment j of the activations = output pixels wm*128 + 16j + fr of the
  // tile (row r / W, column r % W): halo pixel upix0 + (16j / W) * HW + 16j % W
  const int offw0 = (wn * WTN + fr) * 64 + ((fq ^ swz4(fr)) << 4);
  const int upix0 = ((wm * WTM) / W) * HW + (wm * WTM) % W + fr;

  // halo of chunk 0, then the weights of K-steps 0..2
#pragma unroll
  for (int i = 0; i < HL; ++i) stage_h(i, 0);
#pragma unroll
  for (int s = 0; s < PDIST; ++s)
    if (s < nk) stage_w();
  __builtin_amdgcn_sched_barrier(0);
  if (p.bias2) {
    int brow[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) brow[j] = (tile_m * BM + wm * WTM + j * 16 + fr) / (int)p.bias2_rows;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = tile_n * BN + wn * WTN + i * 16 + 4 * fq;
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = *(const f32x4 *)(p.bias2 + (int64_t)brow[j] * p.ldb2 + n) + bias_v[i];
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = bias_v[i];
  }
  // wave-uniform count of this wave's DMA pieces that may stay in flight while "K-step kt+1 has landed"
  auto halo_issued = [&](int it) { return it >= 0 && it % 9 < HL && it / 9 + 1 < cpt ? 1 : 0; };
  auto wait_landed = [&](int kt, int l) {       // l = weight pieces per K-step of this wave half
    int later = min(PDIST - 1, nk - 2 - kt);
    if (later < 0) later = 0;
    const int n = later * l + halo_issued(kt - 1) + halo_issued(kt);
    switch (n) {
      case 0: wait_vm<0>(); break;  case 1: wait_vm<1>(); break;  case 2: wait_vm<2>(); break;  case 3: wait_vm<3>(); break;
      case 4: wait_vm<4>(); break;  case 5: wait_vm<5>(); break;  case 6: wait_vm<6>(); break;  case 7: wait_vm<7>(); break;
      default: wait_vm<8>(); break;
    }
  };
  // K-step 0 (and the halo of chunk 0, issued before it) landed; K-steps 1, 2 may stay in flight
  {
    const int left = min(PDIST - 1, nk - 1);
    if (late) wait_dma<L_LATE>(left); else wait_dma<L_EARLY>(left);
  }
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();

  int read_slot = 0, rc = 0, rt = 0;              // K-step being read: chunk rc, tap rt
  for (int kt = 0; kt < nk; ++kt) {
    // ---- READ phase
    const char *sb = smem + read_slot * B_BYTES;
    read_slot = read_slot + 1 == PSTAGES ? 0 : read_slot + 1;
    const char *sh = smem + RING + (rc & 1) * HALO_BYTES;
    const int ky = rt / 3, kx = rt - ky * 3;
    const int tapoff = ky * HW + kx;
    f16x8 fw[TN], fa[TM];
#pragma unroll
    for (int i = 0; i < TN; ++i) fw[i] = *(const f16x8 *)(sb + offw0 + i * 1024);
    const int hp0 = upix0 + tapoff;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int hp = hp0 + ((16 * j) / W) * HW + (16 * j) % W;
      fa[j] = *(const f16x8 *)(sh + (hp << 6) + ((((hp >> 2) & 3) ^ fq) << 4));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (rt < HL && rc + 1 < cpt) stage_h(rt, (rc + 1) & 1);
    if (kt + PDIST < nk) stage_w();
    if (late) wait_landed(kt, L_LATE);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- COMPUTE phase
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if (!late) wait_landed(kt, L_EARLY);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (++rt == 9) { rt = 0; ++rc; }
  }
  if (!late) __builtin_amdgcn_s_barrier();
  pp_epilogue<BM, BN, TN, TM, WTN, WTM, false, NT>(p, acc, smem, tile_m, tile_n, wm, wn, tid, fr, fq);
}

template <int BN, int W>
int launch_halo_t(GemmArgs &a, hipStream_t s) {
  constexpr int HL = ((256 / W + 2) * (W + 2) + 127) / 128;
  constexpr size_t ring = (size_t)4 * BN * 64 + 2 * (size_t)HL * 8 * 1024, tile = (size_t)256 * BN * 2;
  constexpr size_t lds = ring > tile ? ring : tile;
  static_assert(lds <= 160 * 1024, "LDS per workgroup");
  static bool attr_set[SP_MAX_DEVICES] = {};
  if (int rc = sp_ensure_dyn_lds((const void *)gemm_halo_kernel<BN, W>, (int)lds, attr_set, "sp_gemm_f16(halo)")) return rc;
  a.tiles_m = a.m / 256;
  a.tiles_n = a.n / BN;
  note_kernel("gemm_halo_kernel<%d, %d>", BN, W);
  SP_CLEAR_STALE_ERROR();
  hipLaunchKernelGGL((gemm_halo_kernel<BN, W>), dim3(a.tiles_m * a.tiles_n), dim3(512), lds, s, a);
  SP_CHECK_LAUNCH("sp_gemm_f16(halo)");
  return SP_OK;
}

// 3x3 convolutions the halo-tile kernel takes: stride 1, no upsample, whole image rows of 64 or 128 pixels per tile, no
// LayerNorm fold / GEGLU / split-K / Euler tail (their epilogue paths live in the other kernels)
bool halo_supported(const GemmArgs &a) {
  return a.mode == SP_A_CONV3X3 && a.stride == 1 && a.ups == 0 && (a.win == 64 || a.win == 128) && a.hout == a.hin &&
         a.wout == a.win && ((int64_t)a.hin * a.win) % 256 == 0 && a.m % 256 == 0 && a.cin % 32 == 0 && a.cin >= 64 &&
         (a.n % 320 == 0 || a.n % 256 == 0) && !a.geglu && !a.ln_stats && a.ksplit <= 1 && !a.eul_out && a.n_store == 0;
}
int launch_halo(GemmArgs &a, hipStream_t s) {
  const bool n320 = a.n % 320 == 0 && a.n % 256 != 0;
  if (a.win == 128) return n320 ? launch_halo_t<320, 128>(a, s) : launch_halo_t<256, 128>(a, s);
  return n320 ? launch_halo_t<320, 64>(a, s) : launch_halo_t<256, 64>(a, s);
}


/* The test that passed (tests/test_kernels_gpu.py at the time):
@pytest.mark.parametrize("nimg,hh,ww,cin,n,extras", [(2, 8, 128, 320, 320, ""), (3, 4, 128, 64, 640, "b2"), (1, 4, 64, 640, 640, "r"),
                                                      (2, 8, 64, 128, 256, "b2r"), (1, 72, 128, 320, 320, "b2r"), (2, 36, 64, 640, 1280, ""),
                                                      (5, 2, 128, 960, 320, "r")])
def test_gemm_conv3x3_halo_tile(...):  route 5 vs F.conv2d (rel-L2 <= 2e-3) and vs route 2 (<= 1e-3), guard rows intact.
*/

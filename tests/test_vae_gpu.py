"""Temporal VAE decoder on the last stage: HIP engine vs the fp32 oracle restatement of diffusers'
AutoencoderKLTemporalDecoder (oracle/vae_temporal_decoder_ref.py, parity unpinned -- see its header), through the C ABI.
Reference call site: /root/reference/scripts/generate_video_demo.py:154-195."""

import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _ops():
    from vdpp_amd.hip import ops
    return ops


@pytest.mark.parametrize("rows,cols,ld", [(64, 128, 128), (300, 576, 640), (128, 2304, 2304), (40, 9216, 9216),
                                          (7, 16384, 16384), (9, 8, 8)])
def test_softmax_rows(rows, cols, ld):
    """In-place row softmax of fp16 scores (fp32 statistics) vs torch; columns behind `cols` (row pitch ld) untouched."""
    ops = _ops()
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(rows, ld, generator=g) * 4.0).half()
    x[0, :cols] = -60000.0                                  # a row of equal, hugely negative scores: uniform, no NaN
    if cols > 8:
        x[1, 3] = 30000.0                                   # one dominant score
    xd = x.to(DEV)
    ops.softmax_rows(xd, rows=rows, cols=cols, ld=ld)
    torch.cuda.synchronize()
    got = xd.float().cpu()
    want = torch.softmax(x[:, :cols].float(), dim=-1)
    assert torch.isfinite(got).all()
    assert float((got[:, :cols] - want).abs().max()) <= 1e-3 and rel_l2(got[:, :cols], want) <= 2e-3
    assert float((got[:, :cols].sum(-1) - 1).abs().max()) <= 4e-3
    assert torch.equal(got[:, cols:], x[:, cols:].float())


def test_softmax_rows_survives_scores_that_overflowed_fp16():
    """A trained VAE's mid-block scores can leave fp16's range (the reference upcasts the VAE to fp32 for this,
    generate_video_demo.py:171-175); a +-inf that the contraction wrote must not turn the whole row into NaN: it is read
    as +-65504, so one overflowed score takes all of the row's mass and -inf none."""
    ops = _ops()
    x = torch.randn(6, 512).mul(3).half()
    x[1, 7] = float("inf")
    x[2, 100] = float("-inf")
    x[3, 5], x[3, 300] = float("inf"), float("-inf")
    x[4, :] = float("-inf")
    xd = x.to(DEV)
    ops.softmax_rows(xd, rows=6, cols=512)
    torch.cuda.synchronize()
    got = xd.float().cpu()
    assert torch.isfinite(got).all()
    assert float(got[1, 7]) == 1.0 and float(got[1].sum()) == 1.0
    assert float(got[2, 100]) == 0.0 and abs(float(got[2].sum()) - 1) < 4e-3
    assert float(got[3, 5]) == 1.0 and float(got[3, 300]) == 0.0
    assert float((got[4] - 1 / 512).abs().max()) < 1e-5                    # all equal -> uniform
    assert rel_l2(got[0], torch.softmax(x[0].float(), -1)) <= 2e-3 and rel_l2(got[5], torch.softmax(x[5].float(), -1)) <= 2e-3


def test_softmax_rows_keeps_a_nan_a_nan():
    """The saturation above clamps infinities only: a NaN score (an upstream inf - inf) must still poison its row, so that
    decode_latents(check_finite=True) can see the fault."""
    ops = _ops()
    x = torch.randn(3, 256).half()
    x[1, 17] = float("nan")
    xd = x.to(DEV)
    ops.softmax_rows(xd, rows=3, cols=256)
    got = xd.float().cpu()
    assert torch.isfinite(got[0]).all() and torch.isfinite(got[2]).all()
    assert torch.isnan(got[1]).any()


@pytest.mark.parametrize("m,n,k,lda", [(512, 256, 64, None), (300, 512, 512, None), (1000, 768, 128, 192)])
def test_gemm_f32out_is_the_raw_fp32_product(m, n, k, lda):
    """sp_gemm_f32out_f16: D = A W^T as raw fp32 sums (no rounding to fp16 anywhere), ragged m, A with a row pitch."""
    ops = _ops()
    g = torch.Generator().manual_seed(m + n)
    a = torch.randn(m, lda or k, generator=g).half()
    w = torch.randn(n, k, generator=g).half()
    out = torch.full((m, n), -7.0, dtype=torch.float32, device=DEV)
    ops.gemm_f32out(a.to(DEV)[:, :k], w.to(DEV), out, m=m, n=n, k=k, lda=lda or k)
    want = a[:, :k].double() @ w.double().t()
    err = (out.double().cpu() - want).abs().max() / want.abs().max()
    assert float(err) <= 1e-5, f"fp32 product off: {float(err):.2e}"            # fp32 accumulation of exact fp16 products
    with pytest.raises(ops.HipKernelError, match="multiple of 256"):
        ops.gemm_f32out(a.to(DEV)[:, :k], w.to(DEV)[:100], out, m=m, n=100, k=k, lda=lda or k)


@pytest.mark.parametrize("rows,cols,inplace", [(5, 256, True), (9, 2304, True), (4, 9216, True), (6, 1024, False)])
def test_softmax_rows_f32(rows, cols, inplace):
    """fp32 logits -> fp16 probabilities, also IN PLACE over the front of each row's logits (ldo = 2 * ld); logits far
    outside fp16's range (+-3e5) and logits whose differences fp16 could not hold near 60,000 are handled exactly."""
    ops = _ops()
    g = torch.Generator().manual_seed(rows * cols)
    x = torch.randn(rows, cols, generator=g) * 50.0
    x[0] = x[0] * 6000.0                                     # |logit| up to ~1e6
    x[1] = 60000.0 + torch.randn(cols, generator=g) * 3.0    # fp16 spacing at 60,000 is 32: the differences would vanish
    x[2, 5] = 2.0e5                                          # one dominant logit beyond fp16's largest number
    scale = 0.37
    xd = x.to(DEV)
    if inplace:
        out = xd.view(torch.float16)
        ops.softmax_rows_f32(xd, out, rows=rows, cols=cols, scale=scale, ld=cols, ldo=2 * cols)
        got = out[:, :cols].float().cpu()
    else:
        out = torch.full((rows, cols + 8), 3.0, dtype=torch.float16, device=DEV)
        ops.softmax_rows_f32(xd, out, rows=rows, cols=cols, scale=scale)
        got = out[:, :cols].float().cpu()
        assert float((out[:, cols:].float() - 3.0).abs().max()) == 0.0 and torch.equal(xd.cpu(), x)
    want = torch.softmax(x.double() * scale, dim=-1).float()
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) <= 1e-3 and rel_l2(got, want) <= 2e-3
    assert float(got[2, 5]) == 1.0
    with pytest.raises(ops.HipKernelError, match="in place"):
        ops.softmax_rows_f32(xd, xd.view(torch.float16), rows=rows, cols=cols, ld=cols, ldo=cols)


def test_mid_block_attention_with_logits_beyond_fp16():
    """The decoder's single-head attention on inputs whose logits reach ~+-1.4e5 -- past fp16's 65,504, with rival keys
    only a few units apart: what trained weights can produce and why the reference upcasts this VAE to fp32
    (scripts/generate_video_demo.py:171-175).  With fp32 logits (round 5) the block matches an fp64 evaluation to 2e-2;
    the fp16-score composition (VDPP_VAE_FP16_SCORES=1 / fp32_scores=False) can only saturate and must be visibly wrong
    on the same input, which is what makes this a test of the new path."""
    from vdpp_amd.models.vae_hip import TemporalDecoderHIP, VAEDecoderConfig, random_state_dict
    cfg = VAEDecoderConfig.tiny(64)
    sd = random_state_dict(cfg, seed=5)
    dec = TemporalDecoderHIP(cfg, sd, DEV)
    p = dec.mid[1]
    c, n_img, hw = p["c"], 2, 256
    g = torch.Generator().manual_seed(9)
    x = torch.randn(n_img * hw, c, generator=g).half()
    # blow the q / k projections up so that q.k / sqrt(c) spans +-4e5 while q, k themselves stay inside fp16
    for name, f in (("q", 300.0), ("k", 300.0)):
        p[name].w = (p[name].w.float() * f).half()
        p[name].bias = p[name].bias * f
    assert dec.fp32_scores
    got32 = dec._run_attn(p, x.to(DEV), n_img, hw).float().cpu()
    dec.fp32_scores = False
    got16 = dec._run_attn(p, x.to(DEV), n_img, hw).float().cpu()
    dec.fp32_scores = True
    # fp64 reference of the same block on the same fp16-rounded weights
    xn = F.group_norm(x.double().reshape(n_img, hw, c).permute(0, 2, 1), cfg.norm_groups,
                      p["norm"].g.double().cpu(), p["norm"].b.double().cpu(), eps=p["norm"].eps).permute(0, 2, 1)
    xn = xn.half().double()                                                   # the engine stores the normalised rows in fp16
    q = (xn @ p["q"].w.double().cpu().t() + p["q"].bias.double().cpu()).half().double()
    k = (xn @ p["k"].w.double().cpu().t() + p["k"].bias.double().cpu()).half().double()
    v = xn @ p["wv"].double().cpu().t() + p["bv"].double().cpu()
    logits = q @ k.transpose(1, 2) / (c ** 0.5)
    assert float(logits.abs().max()) > 1e5, "the test input no longer leaves fp16's range (65,504)"
    o = torch.softmax(logits, dim=-1) @ v
    want = (o.reshape(n_img * hw, c).half().double() @ p["out"].w.double().cpu().t() + p["out"].bias.double().cpu()
            + x.double()).float()
    e32, e16 = rel_l2(got32, want), rel_l2(got16, want)
    print(f"mid-block attention, logits up to {float(logits.abs().max()):.3g}: fp32 logits rel-L2 {e32:.2e}, fp16 scores {e16:.2e}")
    assert torch.isfinite(got32).all() and e32 <= 2e-2
    assert e16 > 5 * e32, "the fp16-score path should not be able to follow these logits"


def test_softmax_rows_rejects_bad_shapes():
    ops = _ops()
    x = torch.zeros(4, 24, dtype=torch.float16, device=DEV)
    with pytest.raises(ops.HipKernelError):
        ops.softmax_rows(x, rows=4, cols=20, ld=24)          # not a multiple of 8
    with pytest.raises(ops.HipKernelError):
        ops.softmax_rows(torch.zeros(2, 16392, dtype=torch.float16, device=DEV), rows=2, cols=16392)


@pytest.mark.parametrize("layout", ["bcfhw", "nchw"])
def test_vae_boundary_kernels(layout):
    """pack (1/scaling_factor, channels-last, zero padded) and frames out (time_conv_out + layout, fp16 / fp32) on a
    chunk in the middle of the flattened (batch, frame) list, both tensor layouts."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    b, f, h, w, cpad = 2, 5, 6, 10, 64
    lat = torch.randn(b, 4, f, h, w, generator=g).half()
    src = lat if layout == "bcfhw" else lat.permute(0, 2, 1, 3, 4).reshape(b * f, 4, h, w).contiguous()
    hw = h * w
    st_in = (4 * f * hw, f * hw, hw) if layout == "bcfhw" else (f * 4 * hw, hw, 4 * hw)
    flat0, n = 3, 4                                          # frames 3,4 of video 0 and 0,1 of video 1
    rows = torch.full((n * hw, cpad), 7.0, dtype=torch.float16, device=DEV)
    ops.vae_pack_latent(src.to(DEV), rows, scale=2.5, flat0=flat0, n=n, frames_per_item=f, strides=st_in, h=h, w=w, cpad=cpad)
    torch.cuda.synchronize()
    flat = lat.permute(0, 2, 3, 4, 1).reshape(b * f, hw, 4)[flat0:flat0 + n].reshape(n * hw, 4)
    assert torch.equal(rows[:, :4].cpu(), (flat.float() * 2.5).half()) and float(rows[:, 4:].abs().max()) == 0.0

    x = torch.randn(n * hw, 8, generator=g).half()
    wt, bs = torch.randn(3, 3, 3, generator=g), torch.randn(3, generator=g)
    x5 = x[:, :3].float().reshape(1, n, h, w, 3).permute(0, 4, 1, 2, 3)             # (1, 3, n, h, w): ONE item of n frames
    want = F.conv3d(x5, wt[:, :, :, None, None], bs, padding=(1, 0, 0))[0].permute(1, 0, 2, 3)   # (n, 3, h, w)
    for dtype in (torch.float16, torch.float32):
        shape = (b, 3, f, h, w) if layout == "bcfhw" else (b * f, 3, h, w)
        out = torch.full(shape, -5.0, dtype=dtype, device=DEV)
        st_out = (3 * f * hw, f * hw, hw) if layout == "bcfhw" else (f * 3 * hw, hw, 3 * hw)
        ops.vae_frames_out(x.to(DEV), wt.to(DEV), bs.to(DEV), out, batch=1, frames=n, h=h, w=w, flat0=flat0,
                           frames_per_item=f, strides=st_out)
        torch.cuda.synchronize()
        o = out.float().cpu()
        o = o.permute(0, 2, 1, 3, 4).reshape(b * f, 3, h, w) if layout == "bcfhw" else o
        assert rel_l2(o[flat0:flat0 + n], want) <= (1e-3 if dtype == torch.float16 else 1e-6)
        assert torch.all(o[:flat0] == -5.0) and torch.all(o[flat0 + n:] == -5.0)          # other entries untouched


def _pair(cfg_name, seed):
    from oracle.vae_temporal_decoder_ref import TemporalDecoderRef
    from oracle.vae_temporal_decoder_ref import VAEDecoderConfig as RefCfg
    from vdpp_amd.models.vae_hip import TemporalDecoderHIP, VAEDecoderConfig, random_state_dict

    cfg = VAEDecoderConfig.tiny(64) if cfg_name == "tiny" else VAEDecoderConfig.svd()
    rcfg = RefCfg.tiny(64) if cfg_name == "tiny" else RefCfg.svd()
    sd = random_state_dict(cfg, seed=seed)
    hip = TemporalDecoderHIP(cfg, sd, DEV)
    ref = TemporalDecoderRef(rcfg).eval()
    ref.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    return hip, ref


def test_decoder_matches_oracle_reduced_width():
    """Whole decoder, narrow channels (64..256), 3 frames of an 8 x 8 latent (64 tokens per frame) -> 64 x 64 frames:
    ``decode`` (one call) and ``decode_latents`` (scaling factor, fp32 video tensor, chunks of 14 = one call, and of 2 =
    calls that straddle the two videos of the batch) against the oracle's same functions.  rel-L2 <= 2e-2."""
    from oracle.vae_temporal_decoder_ref import decode_latents
    hip, ref = _pair("tiny", 5)
    g = torch.Generator().manual_seed(11)
    z = torch.randn(3, 4, 8, 8, generator=g).half()
    got = hip.decode(z.to(DEV), 3)
    torch.cuda.synchronize()
    with torch.no_grad():
        want = ref(z.float(), 3)
    assert got.shape == (3, 3, 64, 64) and got.dtype == torch.float16 and torch.isfinite(got).all()
    assert rel_l2(got.float().cpu(), want) <= 2e-2

    lat = (torch.randn(2, 4, 3, 8, 8, generator=g) * 0.18215).half()
    for chunk in (14, 2):
        vid = hip.decode_latents(lat.to(DEV), 3, decode_chunk_size=chunk)
        torch.cuda.synchronize()
        want = decode_latents(lat.float(), ref, 3, decode_chunk_size=chunk)
        assert vid.shape == (2, 3, 3, 64, 64) and vid.dtype == torch.float32
        assert rel_l2(vid.cpu(), want) <= 2e-2, chunk


def test_decoder_argument_checks():
    hip, _ = _pair("tiny", 1)
    with pytest.raises(ValueError):
        hip.decode(torch.zeros(5, 4, 8, 8, dtype=torch.float16, device=DEV), 3)          # 5 frames, 3 per item
    with pytest.raises(ValueError):
        hip.decode(torch.zeros(3, 4, 6, 6, dtype=torch.float16, device=DEV), 3)          # 36 tokens: not a multiple of 64
    with pytest.raises(TypeError):
        hip.decode(torch.zeros(3, 4, 8, 8, dtype=torch.float32, device=DEV), 3)
    with pytest.raises(ValueError):
        hip.decode_latents(torch.zeros(1, 4, 3, 8, 8, dtype=torch.float16, device=DEV), 4)


def test_decoder_full_width_matches_oracle():
    """The real SVD decoder widths (128/256/512/512, 63.6 M parameters, one 512-wide attention head), 2 frames of a
    24 x 32 latent (768 tokens per frame) -> 192 x 256 frames; rel-L2 <= 2e-2 against the fp32 oracle."""
    hip, ref = _pair("svd", 2)
    g = torch.Generator().manual_seed(21)
    z = (torch.randn(2, 4, 24, 32, generator=g) * 4.0).half()
    got = hip.decode(z.to(DEV), 2)
    torch.cuda.synchronize()
    with torch.no_grad():
        want = ref(z.float(), 2)
    assert torch.isfinite(got).all()
    err = rel_l2(got.float().cpu(), want)
    assert err <= 2e-2, f"full-width decoder rel_l2={err:.3e}"


def test_decoder_benchmark_resolution_matches_oracle():
    """The demo's own resolution against the oracle: 72 x 128 latent -> 576 x 1024 frames, 3 frames (1.77 M rows per
    activation at the last level; 21 TFLOP of fp32 for the oracle on the host cores, ~40 s).  The demo's full 14 frames
    are covered by the size-independent test below."""
    hip, ref = _pair("svd", 3)
    g = torch.Generator().manual_seed(31)
    z = (torch.randn(3, 4, 72, 128, generator=g) * 4.0).half()
    got = hip.decode(z.to(DEV), 3)
    torch.cuda.synchronize()
    with torch.no_grad():
        want = ref(z.float(), 3)
    err = rel_l2(got.float().cpu(), want)
    assert torch.isfinite(got).all() and err <= 2e-2, f"benchmark-resolution decoder rel_l2={err:.3e}"


def test_decoder_benchmark_shape_two_kernel_routes_agree_and_are_deterministic():
    """The demo's own decode at FULL size -- 14 frames, 72 x 128 latent -> 576 x 1024 frames, 8.26 M rows per activation at
    the last level, 2.1 GB tensors whose byte offsets end just under 2^31 -- through a size-independent property (the
    oracle's 97 TFLOP of fp32 took 190 of the suite's seconds): the decode through the large-tile ping-pong / persistent
    contraction kernels must agree with the same decode through the independent 128 x 128 / 64 x 64 kernels (different
    tiling, LDS image, pipeline and row addressing), and two launches must be bit-identical.  The oracle pins the same
    weights at this resolution (3 frames, above) and at every width (tests above)."""
    from vdpp_amd.hip import ops
    hip, _ = _pair("svd", 3)
    g = torch.Generator().manual_seed(31)
    z = (torch.randn(14, 4, 72, 128, generator=g) * 4.0).half().to(DEV)
    a = hip.decode(z, 14)
    b = hip.decode(z, 14)
    torch.cuda.synchronize()
    assert a.shape == (14, 3, 576, 1024) and torch.isfinite(a).all()
    assert torch.equal(a, b), "two decodes of the same latent differ"
    del b
    with ops.gemm_route(1):                          # small tiles only: neither gemm_pp.hip nor gemm_ps.hip
        c = hip.decode(z, 14)
    torch.cuda.synchronize()
    err = rel_l2(c.float().cpu(), a.float().cpu())
    assert err <= 5e-3, f"kernel routes disagree at the demo's decode size: rel_l2={err:.3e}"
    # every frame, every image row carries signal (an addressing slip past 2^31 would leave zeros or repeats at the far end)
    assert float(a[-1, :, -8:, :].float().abs().mean()) > 1e-3 and not torch.equal(a[-1], a[-2])


# ------------------------------------------------------------------------------------------------ encoder half
def _enc_pair(cfg_name, seed):
    from oracle.vae_temporal_decoder_ref import EncoderRef
    from oracle.vae_temporal_decoder_ref import VAEDecoderConfig as RefCfg
    from vdpp_amd.models.vae_hip import ImageEncoderHIP, VAEDecoderConfig, random_encoder_state_dict

    cfg = VAEDecoderConfig.tiny(64) if cfg_name == "tiny" else VAEDecoderConfig.svd()
    rcfg = RefCfg.tiny(64) if cfg_name == "tiny" else RefCfg.svd()
    sd = random_encoder_state_dict(cfg, seed=seed)
    hip = ImageEncoderHIP(cfg, sd, DEV)
    ref = EncoderRef(rcfg).eval()
    ref.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    return hip, ref


def test_encoder_boundary_kernels_mirror_round_trip():
    """pack with the mirror, unpack with the mirror: the identity (the encoder engine runs on the mirrored image)."""
    ops = _ops()
    img = torch.randn(2, 3, 6, 10, generator=torch.Generator().manual_seed(4)).half()
    for flip in (False, True):
        rows = torch.full((2 * 60, 64), 3.0, dtype=torch.float16, device=DEV)
        ops.vae_image_pack(img.to(DEV), rows, batch=2, h=6, w=10, cpad=64, flip=flip)
        want = img.flip(-1, -2) if flip else img
        assert torch.equal(rows[:, :3].cpu(), want.permute(0, 2, 3, 1).reshape(120, 3)) and float(rows[:, 3:].abs().max()) == 0
        out = torch.zeros(2, 3, 4, 6, 10, dtype=torch.float16, device=DEV)
        ops.vae_latent_out(rows, out, batch=2, channels=3, frames=4, h=6, w=10, flip=flip)
        assert torch.equal(out.cpu(), img[:, :, None].expand(2, 3, 4, 6, 10))


@pytest.mark.parametrize("cfg_name,shape", [("tiny", (2, 3, 64, 64)), ("tiny", (1, 3, 64, 128)), ("svd", (1, 3, 128, 256))])
def test_encoder_matches_oracle(cfg_name, shape):
    """``vae.encode(image).latent_dist.mode()`` repeated over the frames (ref generate_video_demo.py:139-148): the 2-D
    encoder with its bottom/right-padded stride-2 convolutions, the single-head mid attention and quant_conv; narrow and
    real widths (34.2 M parameters); rel-L2 <= 2e-2 against the fp32 oracle (parity unpinned, see its header)."""
    from oracle.vae_temporal_decoder_ref import encode_image_latents
    hip, ref = _enc_pair(cfg_name, 7)
    img = torch.randn(*shape, generator=torch.Generator().manual_seed(13)).clamp(-1.5, 1.5).half()
    got = hip.encode_image_latents(img.to(DEV), 3)
    torch.cuda.synchronize()
    want = encode_image_latents(img.float(), ref, 3)
    assert got.shape == want.shape and got.dtype == torch.float16 and torch.isfinite(got).all()
    assert rel_l2(got.float().cpu(), want) <= 2e-2
    with pytest.raises(ValueError):
        hip.encode_image_latents(torch.zeros(1, 3, 60, 64, dtype=torch.float16, device=DEV), 3)


def test_image_to_video_flow_through_the_edge_stages():
    """The reference demo's data flow at toy size (scripts/generate_video_demo.py: encode_image -> set_conditioning ->
    denoising steps through the pipeline executor -> decode_latents), every stage on its HIP engine: shapes, dtypes and
    finiteness at each boundary, and the edge stages equal their oracles on the very tensors that flowed through."""
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    from oracle.vae_temporal_decoder_ref import decode_latents as ref_decode, encode_image_latents
    from vdpp_amd.models.clip_hip import CLIPVisionHIP, CLIPVisionSpec
    from vdpp_amd.models.edge_stages import decode_latents, encode_image
    from vdpp_amd.models.svd_unet import StableVideoUNet
    from vdpp_amd.models.unet_hip import SVDUNetHIP
    from vdpp_amd.models.unet_spec import UNetConfig, random_state_dict
    from vdpp_amd.pipeline import LatentSpec, run_single_latent

    frames, h, w, steps = 3, 8, 16, 2
    ucfg = UNetConfig.tiny(64)                                         # cross_attention_dim 128
    ccfg = CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, image_size=56,
                            patch_size=14, projection_dim=ucfg.cross_attention_dim)
    torch.manual_seed(3)
    clip_ref = CLIPVisionModelWithProjection(ccfg).eval()
    with torch.no_grad():
        for p in clip_ref.parameters():
            p.copy_(p.half().float())
    clip = CLIPVisionHIP(CLIPVisionSpec.from_config(ccfg), {k: v.half() for k, v in clip_ref.state_dict().items()}, DEV)
    enc, enc_ref = _enc_pair("tiny", 17)
    dec, dec_ref = _pair("tiny", 19)
    g = torch.Generator().manual_seed(23)
    pixel_values = torch.randn(1, 3, 56, 56, generator=g).half()
    image = torch.randn(1, 3, 8 * h, 8 * w, generator=g).clamp(-1, 1).half()
    noise = torch.randn(1, 3, 8 * h, 8 * w, generator=g).half()

    emb, img_lat = encode_image(pixel_values, image, clip, enc, frames, noise=noise, noise_aug_strength=0.02)
    assert emb.shape == (1, 1, ucfg.cross_attention_dim) and img_lat.shape == (1, 4, frames, h, w)
    with torch.no_grad():
        assert rel_l2(emb.float().cpu(), clip_ref(pixel_values.float()).image_embeds.unsqueeze(1)) <= 1e-2
    noisy = (image.float() + 0.02 * noise.float()).half().float()
    assert rel_l2(img_lat.float().cpu(), encode_image_latents(noisy, enc_ref, frames)) <= 2e-2

    unet = SVDUNetHIP(ucfg, random_state_dict(ucfg, seed=0, dtype=torch.float16), DEV)
    model = StableVideoUNet(unet=unet, timesteps=StableVideoUNet._default_timestep_schedule(steps))
    model.set_conditioning(emb, img_lat, num_frames=frames)
    lat = (torch.randn(1, 4, frames, h, w, generator=g) * model.init_noise_sigma).half().to(DEV)
    spec = LatentSpec(shape=lat.shape, dtype=torch.float16, device=torch.device(DEV))
    out = run_single_latent(model, total_steps=steps, timesteps=list(range(steps)), world_size=1, rank=0, latent_spec=spec,
                            input_latent=lat)
    assert out.shape == lat.shape and torch.isfinite(out).all()

    video = decode_latents(out, dec, frames)
    torch.cuda.synchronize()
    assert video.shape == (1, 3, frames, 8 * h, 8 * w) and video.dtype == torch.float32 and torch.isfinite(video).all()
    assert rel_l2(video.cpu(), ref_decode(out.float().cpu(), dec_ref, frames)) <= 2e-2


def test_edge_engines_load_from_a_local_checkpoint_directory(tmp_path):
    """ref scripts/generate_video_demo.py:248-262 loads the CLIP image encoder and the VAE by model id; with this backend a
    LOCAL directory in the hub layout (image_encoder/, vae/: config.json + *.safetensors) gives the three engines, which
    must compute what engines built from the same state_dicts compute; a model NAME is refused (no network)."""
    import json

    from safetensors.torch import save_file
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    from vdpp_amd.models.clip_hip import CLIPVisionHIP, CLIPVisionSpec
    from vdpp_amd.models.edge_stages import load_edge_engines
    from vdpp_amd.models.vae_hip import (ImageEncoderHIP, TemporalDecoderHIP, VAEDecoderConfig, random_encoder_state_dict,
                                         random_state_dict)

    ccfg = CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, image_size=56,
                            patch_size=14, projection_dim=64)
    torch.manual_seed(5)
    csd = {k: v.half().contiguous() for k, v in CLIPVisionModelWithProjection(ccfg).state_dict().items()}
    vcfg = VAEDecoderConfig.tiny(64)
    dsd, esd = random_state_dict(vcfg, seed=7), random_encoder_state_dict(vcfg, seed=8)
    (tmp_path / "image_encoder").mkdir(); (tmp_path / "vae").mkdir()
    save_file(csd, str(tmp_path / "image_encoder" / "model.fp16.safetensors"))
    save_file({k: torch.zeros_like(v) for k, v in csd.items()}, str(tmp_path / "image_encoder" / "model.safetensors"))  # ignored
    (tmp_path / "image_encoder" / "config.json").write_text(json.dumps(ccfg.to_dict()))
    save_file({**{"decoder." + k: v.contiguous() for k, v in dsd.items()}, **{k: v.contiguous() for k, v in esd.items()}},
              str(tmp_path / "vae" / "diffusion_pytorch_model.safetensors"))
    (tmp_path / "vae" / "config.json").write_text(json.dumps({"block_out_channels": list(vcfg.block_out_channels),
                                                               "latent_channels": 4, "layers_per_block": 2,
                                                               "scaling_factor": vcfg.scaling_factor}))
    clip, enc, dec = load_edge_engines(str(tmp_path), DEV)
    g = torch.Generator().manual_seed(3)
    px = torch.randn(1, 3, 56, 56, generator=g).half().to(DEV)
    img = torch.randn(1, 3, 64, 64, generator=g).clamp(-1, 1).half().to(DEV)
    lat = torch.randn(1, 4, 2, 8, 8, generator=g).half().to(DEV)
    assert torch.equal(clip(px), CLIPVisionHIP(CLIPVisionSpec.from_config(ccfg), csd, DEV)(px))
    assert torch.equal(enc.encode_image_latents(img, 2), ImageEncoderHIP(vcfg, esd, DEV).encode_image_latents(img, 2))
    assert torch.equal(dec.decode_latents(lat, 2), TemporalDecoderHIP(vcfg, dsd, DEV).decode_latents(lat, 2))
    with pytest.raises(ValueError, match="LOCAL checkpoint directory"):
        load_edge_engines("stabilityai/stable-video-diffusion-img2vid-xt", DEV)

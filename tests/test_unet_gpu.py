"""Whole-UNet and whole-step parity on the GPU: HIP engine vs the fp32 oracle restatement
(``oracle/svd_unet_ref.py``) on identical (fp16-rounded) weights and inputs.

Tolerance: fp16 storage between ~700 kernels vs an fp32 CPU evaluation -> relative L2 <= 2e-2 on the
UNet output, <= 2e-2 on the updated latent (stated per test)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_l2(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _build(c=64, seed=3):
    from oracle.svd_unet_ref import SVDUNetConfig, SVDUNetRef
    from vdpp_amd.models.unet_hip import SVDUNetHIP
    from vdpp_amd.models.unet_spec import UNetConfig, random_state_dict

    cfg = UNetConfig.tiny(c)
    sd = random_state_dict(cfg, seed=seed, dtype=torch.float16)
    ref = SVDUNetRef(SVDUNetConfig.tiny(c)).eval()
    ref.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    return cfg, sd, ref, SVDUNetHIP(cfg, sd, DEV)


@pytest.mark.parametrize("frames,h,w", [(3, 16, 24), (14, 8, 16), (25, 8, 8)])
def test_unet_forward_matches_oracle(frames, h, w):
    cfg, sd, ref, hip = _build()
    g = torch.Generator().manual_seed(11)
    sample = torch.randn(1, frames, 8, h, w, generator=g).half()
    ctx = torch.randn(1, 1, cfg.cross_attention_dim, generator=g).half()
    ids = torch.tensor([[5.0, 127.0, 0.02]]).half()
    t = 1.63777
    with torch.no_grad():
        want = ref(sample.float(), t, ctx.float(), ids.float())[0]
    got = hip(sample.to(DEV), t, ctx.to(DEV), ids.to(DEV))[0]
    torch.cuda.synchronize()
    assert got.shape == want.shape
    assert torch.isfinite(got).all()
    err = rel_l2(got.float(), want)
    assert err <= 2e-2, f"UNet forward rel_l2={err:.3e}"


@pytest.mark.parametrize("frames,h,w", [(25, 16, 16), (14, 16, 24)])
def test_unet_forward_fp8_attention_matches_oracle(frames, h, w):
    """BASELINE config 5 path: spatial self-attention on fp8-e4m3 MFMA, everything else fp16.
    Tolerance: rel-L2 <= 3e-2 against the fp32 oracle, and the fp8 engine must really differ from the
    fp16 one (i.e. the option is not silently ignored)."""
    from vdpp_amd.models.unet_hip import SVDUNetHIP
    cfg, sd, ref, hip16 = _build()
    hip8 = SVDUNetHIP(cfg, sd, DEV, fp8_attention=True)
    hip8.FP8_MIN_SEQ = 1          # the test latents are small: force the fp8 kernel at every level
    assert hip8.fp8_attention and not hip16.fp8_attention
    g = torch.Generator().manual_seed(12)
    sample = torch.randn(1, frames, 8, h, w, generator=g).half()
    ctx = torch.randn(1, 1, cfg.cross_attention_dim, generator=g).half()
    ids = torch.tensor([[5.0, 127.0, 0.02]]).half()
    t = 0.73
    with torch.no_grad():
        want = ref(sample.float(), t, ctx.float(), ids.float())[0]
    got8 = hip8(sample.to(DEV), t, ctx.to(DEV), ids.to(DEV))[0].float()
    got16 = hip16(sample.to(DEV), t, ctx.to(DEV), ids.to(DEV))[0].float()
    torch.cuda.synchronize()
    assert torch.isfinite(got8).all()
    e8, e16 = rel_l2(got8, want), rel_l2(got16, want)
    print(f"UNet forward rel-L2 vs oracle: fp16 attention {e16:.3e}, fp8 attention {e8:.3e}")
    assert e8 <= 3e-2, f"fp8-attention UNet forward rel_l2={e8:.3e}"
    assert e16 <= 2e-2 and not torch.equal(got8, got16)


@pytest.mark.parametrize("guidance", [None, 3.0])
def test_step_matches_oracle_step(guidance):
    """StableVideoUNet.forward (HIP) vs oracle svd_step driving the oracle UNet."""
    from oracle.svd_step_ref import svd_step
    from vdpp_amd.models.svd_unet import StableVideoUNet

    cfg, sd, ref, hip = _build(seed=5)
    steps = 25
    model = StableVideoUNet(unet=hip, timesteps=StableVideoUNet._default_timestep_schedule(steps))
    frames, h, w = 4, 8, 16
    g = torch.Generator().manual_seed(21)
    emb = torch.randn(1, 1, cfg.cross_attention_dim, generator=g).half()
    img = torch.randn(1, 4, frames, h, w, generator=g).half()
    model.set_conditioning(emb.to(DEV), img.to(DEV), guidance_scale=guidance, num_frames=frames)
    for step in (0, 12, 24):
        lat = (torch.randn(1, 4, frames, h, w, generator=g) * float(model.sigmas[step] + 1)).half()
        got = model(lat.to(DEV), step)
        with torch.no_grad():
            want = svd_step(ref, lat.float(), step, sigmas=model.sigmas, timesteps=model.scheduler_timesteps,
                            image_embeddings=emb.float(), image_latents=img.float(),
                            added_time_ids=torch.tensor([[5.0, 127.0, 0.02]]).half().float(),
                            guidance_scale=guidance, dtype=torch.float32)
        err = rel_l2(got.float(), want)
        assert err <= 2e-2, f"step {step} guidance {guidance}: rel_l2={err:.3e}"


def _per_video_oracle(ref, sample, t, ctx, ids):
    """The fp32 oracle on every video of a batch SEPARATELY (batch-1 calls, the configuration the other oracle tests pin),
    so that the comparison does not lean on the oracle's own batch handling; the oracle's batched call must agree."""
    with torch.no_grad():
        each = [ref(sample[i:i + 1].float(), t, ctx[i:i + 1].float(), ids[i:i + 1].float())[0] for i in range(sample.shape[0])]
        both = ref(sample.float(), t, ctx.float(), ids.float())[0]
    want = torch.cat(each, dim=0)
    assert rel_l2(both, want) <= 1e-5, "the oracle's batched call disagrees with its per-video calls"
    return want


@pytest.mark.parametrize("frames,h,w,fp8", [(14, 8, 16, False), (25, 8, 8, False), (3, 16, 24, False), (14, 16, 24, True),
                                           (25, 16, 16, True)])
def test_unet_forward_batch2_distinct_conditioning_matches_oracle(frames, h, w, fp8):
    """Two videos per UNet call -- what bench.py's headline configuration runs -- with DIFFERENT input latents and
    DIFFERENT encoder_hidden_states per video, each video against its own batch-1 oracle forward.  Batch > 1 has code
    of its own (the per-video cross-attention bias rows, the repeated frame position embedding, the time-embedding row
    blocks, temporal attention with batch = B): a wrong row in any of them moves one video's output by O(1).
    Tolerances as at batch 1: 2e-2 (fp16 attention), 3e-2 (fp8 attention) relative L2 PER VIDEO."""
    from vdpp_amd.models.unet_hip import SVDUNetHIP
    cfg, sd, ref, hip = _build(seed=41)
    if fp8:
        hip = SVDUNetHIP(cfg, sd, DEV, fp8_attention=True)
        hip.FP8_MIN_SEQ = 1
    g = torch.Generator().manual_seed(100 + frames)
    sample = torch.randn(2, frames, 8, h, w, generator=g).half()
    sample[1] *= 1.7                                             # the two videos differ in scale as well as in content
    ctx = torch.randn(2, 1, cfg.cross_attention_dim, generator=g).half()
    ids = torch.tensor([[5.0, 127.0, 0.02]]).half().repeat(2, 1)
    t = 0.91
    want = _per_video_oracle(ref, sample, t, ctx, ids)
    got = hip(sample.to(DEV), t, ctx.to(DEV), ids.to(DEV))[0].float().cpu()
    assert got.shape == want.shape and torch.isfinite(got).all()
    tol = 3e-2 if fp8 else 2e-2
    for i in range(2):
        err = rel_l2(got[i], want[i])
        assert err <= tol, f"video {i} of the pair: rel_l2={err:.3e}"
    # and the pair must not be two copies of one video's result (a batch index dropped somewhere)
    assert rel_l2(got[0], want[1]) > 0.3
    # swapping the two videos' conditioning must swap nothing but the conditioning's effect: video 0 with video 1's
    # context differs from video 0 with its own
    swapped = hip(sample.to(DEV), t, ctx.flip(0).to(DEV), ids.to(DEV))[0].float().cpu()
    assert rel_l2(swapped[0], got[0]) > 1e-3, "encoder_hidden_states of video 1 never reaches video 1's rows"


def test_unet_call_refuses_per_video_time_ids_and_timesteps():
    """The engine evaluates the timestep / added-time embedding once per call and shares it across the batch (the
    reference adapter feeds a scalar timestep and `added_time_ids.repeat(batch, 1)`, ref svd_unet.py:252-259,389-392):
    rows that differ would be silently ignored, so they raise ValueError."""
    cfg, sd, ref, hip = _build(seed=43)
    sample = torch.zeros(2, 2, 8, 8, 8, dtype=torch.float16, device=DEV)
    ctx = torch.zeros(2, 1, cfg.cross_attention_dim, dtype=torch.float16, device=DEV)
    same = torch.tensor([[5.0, 127.0, 0.02]] * 2)
    assert hip(sample, 0.5, ctx, same)[0].shape == (2, 2, 4, 8, 8)
    assert hip(sample, torch.tensor([0.5, 0.5]), ctx, same[:1])[0].shape == (2, 2, 4, 8, 8)
    with pytest.raises(ValueError, match="added_time_ids"):
        hip(sample, 0.5, ctx, torch.tensor([[5.0, 127.0, 0.02], [24.0, 127.0, 0.02]]))
    with pytest.raises(ValueError, match="added_time_ids"):
        hip(sample, 0.5, ctx, torch.zeros(3, 3))
    with pytest.raises(ValueError, match="timestep"):
        hip(sample, torch.tensor([0.5, 0.7]), ctx, same)
    with pytest.raises(ValueError, match="encoder_hidden_states"):
        hip(sample, 0.5, ctx[:1], same)


@pytest.mark.parametrize("guidance,batched_cfg,frames", [(None, False, 14), (3.0, False, 14), (3.0, True, 4), (None, False, 25)])
def test_step_batch2_distinct_conditioning_matches_oracle_step(guidance, batched_cfg, frames):
    """StableVideoUNet.forward on a micro-batch of two videos (bench.py's pipeline sample) with different CLIP
    embeddings, image latents and input latents per video, against the oracle step of EACH video alone; guidance off,
    3.0 sequential and 3.0 batched (UNet batch 4).  Tolerance 2e-2 relative L2 on each video's new latent."""
    from oracle.svd_step_ref import svd_step
    from vdpp_amd.models.svd_unet import StableVideoUNet

    cfg, sd, ref, hip = _build(seed=45)
    steps = 25
    model = StableVideoUNet(unet=hip, timesteps=StableVideoUNet._default_timestep_schedule(steps), batched_cfg=batched_cfg)
    h, w = 8, 16
    g = torch.Generator().manual_seed(77)
    emb = torch.randn(2, 1, cfg.cross_attention_dim, generator=g).half()
    img = torch.randn(2, 4, frames, h, w, generator=g).half()
    img[1] *= 0.5
    model.set_conditioning(emb.to(DEV), img.to(DEV), guidance_scale=guidance, num_frames=frames)
    ids = torch.tensor([[5.0, 127.0, 0.02]]).half().float()
    for step in (0, 13, 24):
        lat = (torch.randn(2, 4, frames, h, w, generator=g) * float(model.sigmas[step] + 1)).half()
        got = model(lat.to(DEV), step).float().cpu()
        for i in range(2):
            with torch.no_grad():
                want = svd_step(ref, lat[i:i + 1].float(), step, sigmas=model.sigmas, timesteps=model.scheduler_timesteps,
                                image_embeddings=emb[i:i + 1].float(), image_latents=img[i:i + 1].float(),
                                added_time_ids=ids, guidance_scale=guidance, dtype=torch.float32)
            err = rel_l2(got[i:i + 1], want)
            assert err <= 2e-2, f"step {step} video {i} guidance {guidance}: rel_l2={err:.3e}"
            # the UPDATE (new - old) is what a wrong conditioning row would move; the latent itself is dominated by sigma*noise
            # (not at the last step: sigma = 0.002 there and the update is below the fp16 spacing of the latent)
            upd_g, upd_w = got[i:i + 1] - lat[i:i + 1].float(), want - lat[i:i + 1].float()
            assert step == 24 or rel_l2(upd_g, upd_w) <= 5e-2, f"step {step} video {i}: update off"


def test_step_arithmetic_matches_reference_golden(golden_dir):
    """pack_input + Euler/CFG kernels vs the reference's own StableVideoUNet.forward outputs
    (tests/golden/svd_step.npz, minted with a stub UNet).  fp16 case, tolerance 2e-3 relative L2."""
    from tests.golden.make_golden import _stub_unet
    from vdpp_amd.hip import ops

    z = np.load(f"{golden_dir}/svd_step.npz")
    sig = z["sigmas"]; ts = z["scheduler_timesteps"]
    emb = torch.from_numpy(z["image_embeddings"]).half().to(DEV)
    img = torch.from_numpy(z["image_latents"]).half().to(DEV)
    ids = torch.from_numpy(z["added_time_ids.fp16"]).half().to(DEV)
    b, _, f, h, w = img.shape
    for gs_name, gscale in (("nocfg", None), ("cfg3", 3.0)):
        for step in (0, 12, 24):
            x = torch.from_numpy(z[f"in.fp16.{gs_name}.{step}"]).half().to(DEV)
            want = torch.from_numpy(z[f"out.fp16.{gs_name}.{step}"])
            sigma, sigma_next = float(sig[step]), float(sig[step + 1])

            def run(img_lat, e):
                rows = torch.empty(b * f * h * w, 8, dtype=torch.float16, device=DEV)
                ops.pack_input(x, img_lat, rows, in_scale=1.0 / (sigma * sigma + 1) ** 0.5, b=b, frames=f, h=h, w=w, cpad=8)
                sample = rows.reshape(b, f, h, w, 8).permute(0, 1, 4, 2, 3)
                out = _stub_unet(sample, float(ts[step]), e, ids)[0]          # (B,F,4,H,W), test-side stub
                return out.permute(0, 1, 3, 4, 2).reshape(-1, 4).contiguous()

            eps_c = run(img, emb)
            eps_u = run(torch.zeros_like(img), torch.zeros_like(emb)) if gscale else None
            gvec = torch.linspace(1.0, gscale, f).half().float().to(DEV) if gscale else None
            new = torch.empty_like(x)
            ops.euler_step(x, eps_c, eps_u, gvec, new, ld_eps=4, sigma=sigma, sigma_next=sigma_next, b=b, frames=f, h=h, w=w)
            err = rel_l2(new.float(), want)
            assert err <= 2e-3, f"{gs_name} step {step}: rel_l2={err:.3e}"


def test_adapter_errors_and_api_surface():
    """Same error behaviour as the reference adapter (svd_unet.py:367-375) and the same public attributes."""
    from vdpp_amd.models.svd_unet import StableVideoUNet

    cfg, sd, ref, hip = _build(seed=9)
    model = StableVideoUNet(unet=hip, timesteps=StableVideoUNet._default_timestep_schedule(25))
    lat = torch.zeros(1, 4, 2, 8, 8, dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError):
        model(lat, 0)                                   # conditioning not set
    model.set_dummy_conditioning(1, 2, 8, 8, torch.device(DEV))
    for bad in (-1, 25, 100):
        with pytest.raises(ValueError):
            model(lat, bad)
    out = model(lat, 24)                                # last step: sigma_next == 0
    assert out.shape == lat.shape and out.dtype == torch.float16 and torch.isfinite(out).all()
    assert abs(model.init_noise_sigma - (700.0 ** 2 + 1) ** 0.5) < 1e-3
    assert len(model.sigmas) == 26 and float(model.sigmas[-1]) == 0.0
    assert model._added_time_ids.tolist() == [[5.0, 127.0, torch.tensor(0.02).half().item()]]
    model.clear_conditioning()
    with pytest.raises(RuntimeError):
        model(lat, 0)
    with pytest.raises(ValueError):
        StableVideoUNet.from_pretrained("stabilityai/stable-video-diffusion-img2vid-xt")   # hub id: no network


def test_adapter_rejects_conditioning_of_another_shape():
    """The kernels get raw pointers + the latent's (B, F, H, W): mismatched conditioning must raise, not read out of
    bounds (the reference fails in torch.cat / on broadcast, svd_unet.py:385-411)."""
    from vdpp_amd.models.svd_unet import StableVideoUNet

    cfg, sd, ref, hip = _build(seed=10)
    model = StableVideoUNet(unet=hip, timesteps=StableVideoUNet._default_timestep_schedule(25))
    dev = torch.device(DEV)
    model.set_dummy_conditioning(1, 14, 8, 8, dev, guidance_scale=3.0)
    ok = torch.zeros(1, 4, 14, 8, 8, dtype=torch.float16, device=DEV)
    assert model(ok, 0).shape == ok.shape
    for shape in ((1, 4, 25, 8, 8),          # conditioning for 14 frames, latent with 25
                  (2, 4, 14, 8, 8),          # conditioning batch smaller than the latent batch
                  (1, 4, 14, 16, 8),         # other spatial size
                  (1, 8, 14, 8, 8),          # not 4 latent channels
                  (4, 14, 8, 8)):            # not 5-D
        with pytest.raises(ValueError):
            model(torch.zeros(shape, dtype=torch.float16, device=DEV), 0)
    emb = torch.randn(1, 2, cfg.cross_attention_dim, dtype=torch.float16, device=DEV)     # two context tokens
    model.set_conditioning(emb, torch.zeros_like(ok), num_frames=14)
    with pytest.raises(ValueError):
        model(ok, 0)
    model.set_conditioning(emb[:, :1], torch.zeros_like(ok), guidance_scale=2.0, num_frames=7)   # guidance for 7 frames
    with pytest.raises(ValueError):
        model(ok, 0)


def test_pipeline_stage_on_gpu_single_rank_matches_loop():
    """PipelineStage (world_size 1) driving the HIP adapter == calling the adapter in a loop (bit-identical)."""
    from vdpp_amd.models.svd_unet import StableVideoUNet
    from vdpp_amd.pipeline import LatentSpec, run_single_latent

    cfg, sd, ref, hip = _build(seed=13)
    steps = 5
    model = StableVideoUNet(unet=hip, timesteps=StableVideoUNet._default_timestep_schedule(steps))
    torch.manual_seed(3)
    model.set_dummy_conditioning(1, 3, 8, 8, torch.device(DEV))
    lat = (torch.randn(1, 4, 3, 8, 8) * model.init_noise_sigma).half().to(DEV)
    spec = LatentSpec(shape=lat.shape, dtype=torch.float16, device=torch.device(DEV))
    out = run_single_latent(model, total_steps=steps, timesteps=list(range(steps)), world_size=1, rank=0,
                            latent_spec=spec, input_latent=lat)
    want = lat
    for s in range(steps):
        want = model(want, s)
    assert torch.equal(out, want)


def test_batched_cfg_equals_sequential_cfg():
    """Batch-2 CFG forward (extension) vs the reference's two sequential passes: same kernels on the same rows,
    only GEMM tile scheduling differs -> bit-identical or within fp16 round-off (1e-3 relative L2)."""
    from vdpp_amd.models.svd_unet import StableVideoUNet

    cfg, sd, ref, hip = _build(seed=17)
    ts = StableVideoUNet._default_timestep_schedule(25)
    seq = StableVideoUNet(unet=hip, timesteps=ts)
    bat = StableVideoUNet(unet=hip, timesteps=ts, batched_cfg=True)
    g = torch.Generator().manual_seed(5)
    frames, h, w = 4, 8, 16
    emb = torch.randn(1, 1, cfg.cross_attention_dim, generator=g).half().to(DEV)
    img = torch.randn(1, 4, frames, h, w, generator=g).half().to(DEV)
    for m in (seq, bat):
        m.set_conditioning(emb, img, guidance_scale=3.0, num_frames=frames)
    lat = (torch.randn(1, 4, frames, h, w, generator=g) * 50).half().to(DEV)
    a, b = seq(lat, 7), bat(lat, 7)
    assert rel_l2(b.float(), a.float().cpu()) <= 1e-3


@pytest.mark.parametrize("guidance", [None, 2.5])
def test_euler_tail_in_conv_out_epilogue_is_bit_identical(guidance):
    """the guidance mix + Euler update run in the epilogue of the UNet's last convolution (no eps rows in
    HBM, no separate launch).  Same arithmetic and roundings as sp_euler_step_f16 on stored eps rows -> the new latent
    is bit-identical to the two-kernel path (ref svd_unet.py:410-439)."""
    from vdpp_amd.hip import ops
    from vdpp_amd.models.svd_unet import StableVideoUNet

    cfg, sd, ref, hip = _build(seed=31)
    model = StableVideoUNet(unet=hip, timesteps=StableVideoUNet._default_timestep_schedule(25))
    g = torch.Generator().manual_seed(12)
    frames, h, w = 3, 16, 24
    emb = torch.randn(1, 1, cfg.cross_attention_dim, generator=g).half().to(DEV)
    img = torch.randn(1, 4, frames, h, w, generator=g).half().to(DEV)
    model.set_conditioning(emb, img, guidance_scale=guidance, num_frames=frames)
    lat = (torch.randn(1, 4, frames, h, w, generator=g) * 30).half().to(DEV)
    for step in (0, 11, 24):
        fused = model(lat, step)
        sigma, sigma_next = model._sigma_host[step], model._sigma_host[step + 1]
        in_scale = 1.0 / (sigma * sigma + 1.0) ** 0.5
        eps_u = None
        if guidance:
            eps_u = model._unet_pass(lat, model._uncond_image_latents, model._uncond_embeddings, in_scale, step)
        eps_c = model._unet_pass(lat, model._image_latents, model._image_embeddings, in_scale, step)
        want = torch.empty_like(lat)
        ops.euler_step(lat, eps_c, eps_u, model._guidance32 if guidance else None, want, ld_eps=eps_c.shape[1],
                       sigma=sigma, sigma_next=sigma_next, b=1, frames=frames, h=h, w=w)
        assert torch.equal(fused, want), f"step {step}"


def test_graph_replay_equals_eager():
    """HIP-graph replay of a step (capture once, replay with new latents) is bit-identical to eager launches."""
    from vdpp_amd.models.svd_unet import StableVideoUNet

    cfg, sd, ref, hip = _build(seed=19)
    ts = StableVideoUNet._default_timestep_schedule(25)
    eager = StableVideoUNet(unet=hip, timesteps=ts)
    graphed = StableVideoUNet(unet=hip, timesteps=ts)
    graphed.enable_graphs()
    g = torch.Generator().manual_seed(6)
    frames, h, w = 3, 8, 16
    emb = torch.randn(1, 1, cfg.cross_attention_dim, generator=g).half().to(DEV)
    img = torch.randn(1, 4, frames, h, w, generator=g).half().to(DEV)
    for m in (eager, graphed):
        m.set_conditioning(emb, img, num_frames=frames)
    for trial in range(3):                       # first call captures, later calls replay with different data
        lat = (torch.randn(1, 4, frames, h, w, generator=g) * 100).half().to(DEV)
        for step in (0, 3):
            a, b = eager(lat, step), graphed(lat, step)
            assert torch.equal(a, b), f"trial {trial} step {step}"
    assert len(graphed._graphs) == 2
    graphed.set_conditioning(emb * 2, img, num_frames=frames)     # invalidates the captured graphs
    assert len(graphed._graphs) == 0


def test_graph_replay_with_two_videos_in_flight():
    """HIP graphs + concurrent_samples = 2 (what bench.py runs with VDPP_GRAPHS=1): every lane has its own graph,
    static buffers, pool and scratch, so interleaved replays equal the sequential eager results bit for bit."""
    from vdpp_amd.models.svd_unet import StableVideoUNet
    from vdpp_amd.pipeline import LatentSpec, PipelineConfig, PipelineStage

    cfg, sd, ref, hip = _build(seed=29)
    steps = 4
    ts = StableVideoUNet._default_timestep_schedule(steps)
    eager = StableVideoUNet(unet=hip, timesteps=ts)
    graphed = StableVideoUNet(unet=hip, timesteps=ts)
    graphed.enable_graphs()
    torch.manual_seed(8)
    frames, h, w = 3, 16, 16
    emb = torch.randn(1, 1, cfg.cross_attention_dim).half().to(DEV)
    img = torch.randn(1, 4, frames, h, w).half().to(DEV)
    for m in (eager, graphed):
        m.set_conditioning(emb, img, num_frames=frames)
    shape = torch.Size((1, 4, frames, h, w))
    spec = LatentSpec(shape=shape, dtype=torch.float16, device=torch.device(DEV))
    xs = [(torch.randn(shape) * 20 * (i + 1)).half().to(DEV) for i in range(6)]

    def stage_of(model, conc):
        return PipelineStage(model, PipelineConfig(total_steps=steps, world_size=1, rank=0, timesteps=list(range(steps)),
                                                   latent_spec=spec, concurrent_samples=conc))

    def run(stage):
        out = stage.run_many(len(xs), input_supplier=lambda i: xs[i])
        torch.cuda.synchronize()
        return out

    want = run(stage_of(eager, 1))
    two_lanes = stage_of(graphed, 2)             # one stage = one pair of lane streams
    for trial in range(2):                       # first pass captures (per lane), second pass only replays
        got = run(two_lanes)
        for i, (a, b) in enumerate(zip(want, got)):
            assert torch.equal(a, b), f"trial {trial} sample {i}"
    lanes = {k[0] for k in graphed._graphs}
    assert len(lanes) == 2 and len(graphed._graphs) == 2 * steps       # one graph per (lane, step)


def test_interleaved_run_many_equals_sequential():
    """concurrent_samples > 1 (samples interleaved on separate HIP streams) returns exactly the sequential results."""
    from vdpp_amd.models.svd_unet import StableVideoUNet
    from vdpp_amd.pipeline import LatentSpec, PipelineConfig, PipelineStage

    cfg, sd, ref, hip = _build(seed=23)
    steps = 4
    model = StableVideoUNet(unet=hip, timesteps=StableVideoUNet._default_timestep_schedule(steps))
    torch.manual_seed(4)
    model.set_dummy_conditioning(1, 3, 8, 16, torch.device(DEV))
    shape = torch.Size((1, 4, 3, 8, 16))
    spec = LatentSpec(shape=shape, dtype=torch.float16, device=torch.device(DEV))
    xs = [(torch.randn(shape) * 10 * (i + 1)).half().to(DEV) for i in range(5)]
    outs = {}
    for conc in (1, 2, 3):
        stage = PipelineStage(model, PipelineConfig(total_steps=steps, world_size=1, rank=0, timesteps=list(range(steps)),
                                                    latent_spec=spec, concurrent_samples=conc))
        seen = []
        stage.sample_done_hook = seen.append
        outs[conc] = stage.run_many(5, input_supplier=lambda i: xs[i])
        torch.cuda.synchronize()
        assert sorted(seen) == [0, 1, 2, 3, 4]
    for conc in (2, 3):
        for a, b in zip(outs[1], outs[conc]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("conc,rotate,mb", [(1, False, 1), (2, False, 1), (3, False, 1), (2, True, 1), (2, True, 2), (1, False, 2)])
def test_side_stream_link_with_emulated_stream_ordered_p2p(monkeypatch, conc, rotate, mb):
    """Two pipeline ranks emulated in ONE process on one GPU: torch.distributed isend/irecv are replaced by
    stream-ordered mailbox copies (what RCCL P2P provides: the transfer is ordered after the work already enqueued
    on the issuing stream).  Exercises _SideStreamLink (events, fresh receive buffers, record_stream) and the
    interleaved scheduler exactly as the RCCL path does, and checks the 2-rank result == the 1-rank result.
    mb = 2: every pipeline sample is a micro-batch of two videos (bench.py's default), latent (2,4,F,H,W)."""
    import vdpp_amd.pipeline.pipeline as pl
    from vdpp_amd.models.svd_unet import StableVideoUNet
    from vdpp_amd.pipeline import LatentSpec, PipelineConfig, PipelineStage

    # RCCL's rules for un-batched P2P (tests/p2p_emulation.py): per rank pair ONE queue per side holding both directions,
    # tags ignored, a transfer when a send at one head meets a receive at the other; copies are stream-ordered
    from tests.p2p_emulation import PairFifoTransport
    net = PairFifoTransport(timeout=60)
    net.install(monkeypatch, pl.dist)

    cfg, sd, ref, hip = _build(seed=29)
    steps = 5
    model = StableVideoUNet(unet=hip, timesteps=StableVideoUNet._default_timestep_schedule(steps))
    torch.manual_seed(8)
    model.set_dummy_conditioning(mb, 3, 8, 16, torch.device(DEV))
    shape = torch.Size((mb, 4, 3, 8, 16))
    spec = LatentSpec(shape=shape, dtype=torch.float16, device=torch.device(DEV))
    xs = [(torch.randn(shape) * 20).half().to(DEV) for _ in range(5)]

    def stage(rank, world):
        return PipelineStage(model, PipelineConfig(total_steps=steps, world_size=world, rank=rank,
                                                   timesteps=list(range(steps)), latent_spec=spec, balanced=True,
                                                   async_comm=world > 1, concurrent_samples=conc,
                                                   rotate=rotate and world > 1))

    net.bind(0)
    want = stage(0, 1).run_many(5, input_supplier=lambda i: xs[i])
    r0, r1 = stage(0, 2), stage(1, 2)
    assert (r0.step_range.count, r1.step_range.count) == (3, 2)
    assert r0.run_many(5, input_supplier=lambda i: xs[i]) is None     # rank 0: five sends queued towards rank 1
    net.bind(1)
    got = r1.run_many(5)
    r0.drain(); r1.drain()
    torch.cuda.synchronize()
    assert net.crossed is None and net.idle() and len(got) == 5
    for a, b in zip(want, got):
        assert torch.equal(a, b)


@pytest.mark.parametrize("world,num_samples,conc", [(3, 7, 2), (2, 4, 1), (4, 6, 2)])
def test_ring_schedule_with_emulated_stream_ordered_p2p(monkeypatch, world, num_samples, conc):
    """The ring schedule (PipelineConfig.ring) with its GPU transport: `world` ranks emulated by threads of ONE process
    on one GPU; `isend` / `irecv` are replaced by stream-ordered mailbox copies per directed link (what an RCCL
    send/recv provides).  Exercises the event choreography of `_run_many_ring` (compute lanes -> side stream ->
    compute lanes), the interleaved lanes and the final collection, and checks every sample == the 1-rank result."""
    import threading

    import vdpp_amd.pipeline.pipeline as pl
    from vdpp_amd.models.svd_unet import StableVideoUNet
    from vdpp_amd.pipeline import LatentSpec, PipelineConfig, PipelineStage

    from tests.p2p_emulation import PairFifoTransport
    net = PairFifoTransport(timeout=120)                  # RCCL's pair-FIFO rules; world 2: both directions on ONE queue pair
    net.install(monkeypatch, pl.dist)

    cfg, sd, ref, hip = _build(seed=31)
    steps = 7
    model = StableVideoUNet(unet=hip, timesteps=StableVideoUNet._default_timestep_schedule(steps))
    torch.manual_seed(9)
    model.set_dummy_conditioning(1, 3, 8, 16, torch.device(DEV))
    shape = torch.Size((1, 4, 3, 8, 16))
    spec = LatentSpec(shape=shape, dtype=torch.float16, device=torch.device(DEV))
    xs = [(torch.randn(shape) * 20).half().to(DEV) for _ in range(num_samples)]

    def stage(rank, n, ring):
        return PipelineStage(model, PipelineConfig(total_steps=steps, world_size=n, rank=rank, timesteps=list(range(steps)),
                                                   latent_spec=spec, balanced=True, concurrent_samples=conc, ring=ring))

    want = stage(0, 1, False).run_many(num_samples, input_supplier=lambda i: xs[i])
    torch.cuda.synchronize()
    results, errors = {}, []

    def rank_main(rank):
        try:
            net.bind(rank)
            torch.cuda.set_device(0)
            with torch.no_grad():
                results[rank] = stage(rank, world, True).run_many(num_samples, input_supplier=lambda i: xs[i])
            torch.cuda.synchronize()
        except Exception as exc:       # noqa: BLE001  (reported below, in the main thread)
            errors.append((rank, repr(exc)))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads: t.start()
    for t in threads: t.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads), "a rank is stuck"
    assert net.crossed is None and net.idle()
    if world > 2:
        assert all(len(d) == 1 for d in net.directions_per_pair().values()), "a rank pair carried traffic in both directions"
    assert all(results[r] is None for r in range(world - 1))
    got = results[world - 1]
    assert len(got) == num_samples
    for i, (a, b) in enumerate(zip(want, got)):
        assert torch.equal(a, b), f"sample {i}"


def test_full_width_svd_unet_matches_oracle():
    """The REAL SVD architecture (1.52 B parameters, 320/640/1280 channels, 5/10/20 heads) on a small latent
    (4 frames, 32x32 -> 4,096 token rows at level 0, so the large-tile GEMM kernels run) vs the fp32 oracle.
    Tolerance: relative L2 <= 2e-2 (fp16 storage through ~1,100 kernels vs fp32 CPU)."""
    from oracle.svd_unet_ref import SVDUNetConfig, SVDUNetRef
    from vdpp_amd.models.unet_hip import SVDUNetHIP
    from vdpp_amd.models.unet_spec import UNetConfig, random_state_dict

    cfg = UNetConfig.svd()
    sd = random_state_dict(cfg, seed=1, device=DEV, dtype=torch.float16)
    hip = SVDUNetHIP(cfg, sd, DEV)
    with torch.device("meta"):
        ref = SVDUNetRef(SVDUNetConfig.svd())
    ref = ref.to_empty(device="cpu").eval()
    ref.load_state_dict({k: v.float().cpu() for k, v in sd.items()}, strict=True)
    del sd
    g = torch.Generator().manual_seed(31)
    frames, h, w = 4, 32, 32
    sample = torch.randn(1, frames, 8, h, w, generator=g).half()
    ctx = torch.randn(1, 1, 1024, generator=g).half()
    ids = torch.tensor([[5.0, 127.0, 0.02]]).half()
    got = hip(sample.to(DEV), 0.8, ctx.to(DEV), ids.to(DEV))[0]
    torch.cuda.synchronize()
    with torch.no_grad():
        want = ref(sample.float(), 0.8, ctx.float(), ids.float())[0]
    assert torch.isfinite(got).all()
    err = rel_l2(got.float(), want)
    assert err <= 2e-2, f"full-width UNet rel_l2={err:.3e}"
    # the same network on a PAIR of videos with different inputs and context (two videos per call is what bench.py runs):
    # at 320 / 640 channels and 1,024 / 256 tokens per frame the transformer-entry GroupNorm is folded into per-frame
    # weight copies (B*F = 6 instances), the per-video cross-attention bias rows and the repeated position embedding are
    # live, and the 128x128 / split-K routes see 2x the rows
    frames = 3
    sample = torch.randn(2, frames, 8, h, w, generator=g).half()
    sample[1] *= 1.5
    ctx = torch.randn(2, 1, 1024, generator=g).half()
    got = hip(sample.to(DEV), 0.8, ctx.to(DEV), ids.repeat(2, 1).to(DEV))[0].float().cpu()
    want = _per_video_oracle(ref, sample, 0.8, ctx, ids.repeat(2, 1))
    assert torch.isfinite(got).all()
    for i in range(2):
        err = rel_l2(got[i], want[i])
        assert err <= 2e-2, f"full-width UNet, video {i} of a pair: rel_l2={err:.3e}"
    assert rel_l2(got[0], want[1]) > 0.3


@pytest.mark.parametrize("frames", [14, 25])
def test_benchmark_shape_unet_forward_matches_oracle(frames):
    """The headline workloads themselves -- the real SVD architecture (1.52 B parameters) on the benchmark latents
    (14 frames = BASELINE configs 2-4, 25 frames = config 5 / SVD-XT; 72 x 128: 129,024 / 230,400 token rows at level
    0, every kernel at the shape bench.py times) -- against ONE forward of the fp32 oracle on the host cores (about
    40 / 80 s on the GPU box's 16 cores, ~25 / 45 GB of host memory).
    Tolerance: relative L2 <= 2e-2 (fp16 storage through ~1,100 kernels vs fp32 CPU), as at the reduced sizes."""
    import os

    from oracle.svd_unet_ref import SVDUNetConfig, SVDUNetRef
    from vdpp_amd.models.unet_hip import SVDUNetHIP
    from vdpp_amd.models.unet_spec import UNetConfig, random_state_dict

    cfg = UNetConfig.svd()
    sd = random_state_dict(cfg, seed=2, device=DEV, dtype=torch.float16)
    hip = SVDUNetHIP(cfg, sd, DEV)
    with torch.device("meta"):
        ref = SVDUNetRef(SVDUNetConfig.svd())
    ref = ref.to_empty(device="cpu").eval()
    ref.load_state_dict({k: v.float().cpu() for k, v in sd.items()}, strict=True)
    del sd
    g = torch.Generator().manual_seed(33)
    h, w = 72, 128
    sample = torch.randn(1, frames, 8, h, w, generator=g).half()
    ctx = torch.randn(1, 1, 1024, generator=g).half()
    ids = torch.tensor([[6.0, 127.0, 0.02]]).half()
    got = hip(sample.to(DEV), 1.2, ctx.to(DEV), ids.to(DEV))[0]
    torch.cuda.synchronize()
    got = got.float().cpu()
    del hip
    torch.cuda.empty_cache()
    threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    try:
        with torch.no_grad():
            want = ref(sample.float(), 1.2, ctx.float(), ids.float())[0]
    finally:
        torch.set_num_threads(threads)
    assert torch.isfinite(got).all()
    err = rel_l2(got, want)
    print(f"benchmark-shape UNet forward ({frames} frames) vs fp32 oracle: rel_l2 = {err:.3e}")   # (pytest -s / -rP)
    assert err <= 2e-2, f"benchmark-shape UNet rel_l2={err:.3e}"


def test_benchmark_shape_two_kernel_routes_agree_and_are_deterministic(monkeypatch):
    """A size-independent property at the FULL benchmark size (1.52 B parameters, latent (1,4,14,72,128)), beside the
    oracle check above: one UNet step through the large-tile ping-pong GEMM kernels must agree
    with the same step through the independent 128x128 / 64x64 GEMM kernels (different tiling, LDS image and
    pipeline), and repeated launches must be bit-identical (no atomics anywhere).  Tolerance 5e-3 relative L2
    (both routes round to fp16 at the same places; only fp32 summation order differs)."""
    from vdpp_amd.models.svd_unet import StableVideoUNet

    model = StableVideoUNet.from_random_init(StableVideoUNet._default_timestep_schedule(25), seed=0, device=DEV)
    torch.manual_seed(42)
    model.set_dummy_conditioning(1, 14, 72, 128, torch.device(DEV))
    lat = (torch.randn(1, 4, 14, 72, 128) * model.init_noise_sigma).half().to(DEV)
    assert model.unet.long_attention is True, "the frozen-reference kernel is the default for the level-0 rows"
    model.unet.long_attention = False            # a, b, c: every row through attn_spatial_kernel
    a = model(lat, 0)
    b = model(lat, 0)
    assert torch.isfinite(a).all()
    assert torch.equal(a, b), "two launches of the same step differ"
    from vdpp_amd.hip import ops
    with ops.gemm_route(1):                            # small tiles only: neither gemm_pp.hip nor gemm_ps.hip
        c = model(lat, 0)
    # compare the UPDATE (new - old latent): the latent itself is dominated by the unchanged sigma*noise term
    upd_a, upd_c = (a.float() - lat.float()).cpu(), (c.float() - lat.float()).cpu()
    err = rel_l2(upd_c, upd_a)
    assert err <= 5e-3, f"kernel routes disagree at the benchmark shape: rel_l2={err:.3e}"
    # and with the level-0 attention through the frozen-reference kernel (the default: SVDUNetHIP(long_attention=None)):
    # another softmax reference and summation order, same arithmetic
    model.unet.long_attention = True
    seen = []
    orig = ops.attn_spatial_long
    monkeypatch.setattr(ops, "attn_spatial_long", lambda *a, **k: (seen.append(k["seq"]), orig(*a, **k))[1])
    d = model(lat, 0)
    e = model(lat, 0)
    assert seen and set(seen) == {72 * 128}, f"frozen-reference kernel not on the level-0 rows: {seen}"
    assert torch.equal(d, e), "two launches with long_attention differ"
    err = rel_l2((d.float() - lat.float()).cpu(), upd_a)
    assert err <= 2e-3, f"long_attention changes the step at the benchmark shape: rel_l2={err:.3e}"


@pytest.mark.parametrize("frames,steps", [(14, 25), (25, 30)])
def test_benchmark_shape_pair_step_equals_the_two_single_video_steps(frames, steps):
    """bench.py's headline configuration itself: ONE step on a micro-batch of two videos, latent (2,4,F,72,128), through the
    default kernel routes, with different CLIP embeddings / image latents / input latents per video, must equal the two
    batch-1 steps (which the fp32 oracle pins at this very size: test_benchmark_shape_unet_forward_matches_oracle) on
    the UPDATE each applies.  Same kernels, same roundings; what differs is the tiling of 2x the rows (128x128 tiles and
    split-K at the 4,032-row level, 28 / 50 GroupNorm-fold instances, per-video bias rows) -> tolerance 2e-3 relative L2
    per video, and the pair must be deterministic."""
    from vdpp_amd.models.svd_unet import StableVideoUNet

    ts = StableVideoUNet._default_timestep_schedule(steps)
    model = StableVideoUNet.from_random_init(ts, seed=0, device=DEV)
    g = torch.Generator().manual_seed(frames)
    emb = torch.randn(2, 1, 1024, generator=g).half().to(DEV)
    img = torch.randn(2, 4, frames, 72, 128, generator=g).half().to(DEV)
    step = 2
    lat = (torch.randn(2, 4, frames, 72, 128, generator=g) * float(model.sigmas[step])).half().to(DEV)
    model.set_conditioning(emb, img, num_frames=frames)
    pair = model(lat, step)
    again = model(lat, step)
    assert torch.isfinite(pair).all() and torch.equal(pair, again)
    upd_pair = (pair.float() - lat.float()).cpu()
    for i in range(2):
        model.set_conditioning(emb[i:i + 1], img[i:i + 1], num_frames=frames)
        single = model(lat[i:i + 1].contiguous(), step)
        upd = (single.float() - lat[i:i + 1].float()).cpu()
        err = rel_l2(upd_pair[i:i + 1], upd)
        print(f"{frames} frames, video {i}: pair vs single step, rel-L2 of the update {err:.3e}")
        assert err <= 2e-3, f"video {i} of the pair differs from its batch-1 step: rel_l2={err:.3e}"
    # the two videos' updates are different things (conditioning really is per video)
    assert rel_l2(upd_pair[0], upd_pair[1]) > 0.3


def test_config5_shape_fp8_attention_agrees_with_fp16():
    """BASELINE config 5 at FULL size (SVD-XT: 25 frames, latent (1,4,25,72,128), 30 steps, 1.52 B parameters): the
    CPU oracle is out of reach, so the check is agreement of one diffusion step with spatial attention on fp8-e4m3
    MFMA against the same step on the fp16 kernels (which the oracle pins at reduced size), on the UPDATE the step
    applies.  Tolerance 3e-2 relative L2, plus determinism of the fp8 route."""
    from vdpp_amd.models.svd_unet import StableVideoUNet

    ts = StableVideoUNet._default_timestep_schedule(30)
    m16 = StableVideoUNet.from_random_init(ts, seed=0, device=DEV)
    u8 = type(m16.unet).__new__(type(m16.unet))
    u8.__dict__.update(m16.unet.__dict__)               # same packed weights, no second 3 GB copy
    u8.fp8_attention, u8._fp8_ws, u8._gn_ws = True, {}, None
    m8 = StableVideoUNet(unet=u8, timesteps=ts)
    assert not m16.unet.fp8_attention
    for m in (m16, m8):
        torch.manual_seed(43)
        m.set_dummy_conditioning(1, 25, 72, 128, torch.device(DEV))
    lat = (torch.randn(1, 4, 25, 72, 128) * m16.init_noise_sigma).half().to(DEV)
    a16 = m16(lat, 3)
    a8 = m8(lat, 3)
    b8 = m8(lat, 3)
    assert torch.isfinite(a8).all() and torch.equal(a8, b8)
    upd16, upd8 = (a16.float() - lat.float()).cpu(), (a8.float() - lat.float()).cpu()
    err = rel_l2(upd8, upd16)
    print(f"config-5 step, fp8 vs fp16 attention: rel-L2 of the update {err:.3e}")
    assert 0 < err <= 3e-2, f"fp8 attention route off at the config-5 shape: rel_l2={err:.3e}"


def test_feed_forward_pairs_in_row_chunks_are_bit_identical():
    """SVDUNetHIP.FF_CHUNK_BYTES (VDPP_FF_CHUNK_MB): FF1 -> GEGLU -> FF2 run chunk of rows by chunk of rows, so that the hidden
    activation of a chunk is still in the Infinity Cache when FF2 reads it.  Chunks start on tile boundaries, every tile
    computes what it computed before: the forward must not change by a bit -- including the LayerNorm row statistics
    that the chunked FF2 leaves for the next contraction (t_fi2 -> temporal Q/K/V)."""
    from vdpp_amd.models.unet_hip import SVDUNetHIP
    cfg, sd, ref, hip = _build(c=256, seed=47)               # 256 channels: the row statistics come out of the epilogues
    g = torch.Generator().manual_seed(3)
    sample = torch.randn(2, 3, 8, 16, 16, generator=g).half().to(DEV)
    ctx = torch.randn(2, 1, cfg.cross_attention_dim, generator=g).half().to(DEV)
    ids = torch.tensor([[5.0, 127.0, 0.02]] * 2).to(DEV)
    want = hip(sample, 0.6, ctx, ids)[0]
    calls = []
    orig = SVDUNetHIP._gemm

    def counting(self, r, layer, a, **kw):
        calls.append((layer.geglu, kw.get("m")))
        return orig(self, r, layer, a, **kw)

    hip.FF_CHUNK_BYTES, hip.FF_CHUNK_ROUND = 1 << 18, 256
    try:
        SVDUNetHIP._gemm = counting
        got = hip(sample, 0.6, ctx, ids)[0]
    finally:
        SVDUNetHIP._gemm = orig
        hip.FF_CHUNK_BYTES, hip.FF_CHUNK_ROUND = 0, None
    assert any(geglu and m is not None and m < 2 * 3 * 256 for geglu, m in calls), "no feed-forward was chunked"
    assert torch.equal(got, want)


def test_from_pretrained_local_directory(tmp_path):
    """from_pretrained on a LOCAL diffusers-style directory (unet/config.json + *.safetensors) builds the same
    network as handing the state_dict over directly."""
    import json

    from safetensors.torch import save_file
    from vdpp_amd.models.svd_unet import StableVideoUNet
    from vdpp_amd.models.unet_hip import SVDUNetHIP
    from vdpp_amd.models.unet_spec import UNetConfig, random_state_dict

    cfg = UNetConfig.tiny(64)
    sd = random_state_dict(cfg, seed=2, dtype=torch.float16)
    unet_dir = tmp_path / "svd-tiny" / "unet"
    unet_dir.mkdir(parents=True)
    save_file(sd, str(unet_dir / "diffusion_pytorch_model.safetensors"))
    (unet_dir / "config.json").write_text(json.dumps({
        "_class_name": "UNetSpatioTemporalConditionModel", "in_channels": 8, "out_channels": 4,
        "block_out_channels": list(cfg.block_out_channels), "layers_per_block": 2,
        "num_attention_heads": list(cfg.num_attention_heads), "cross_attention_dim": cfg.cross_attention_dim,
        "addition_time_embed_dim": cfg.addition_time_embed_dim,
        "projection_class_embeddings_input_dim": cfg.projection_class_embeddings_input_dim,
        "down_block_types": ["CrossAttnDownBlockSpatioTemporal"] * 3 + ["DownBlockSpatioTemporal"]}))
    loaded = StableVideoUNet.from_pretrained(str(tmp_path / "svd-tiny"), device=DEV)
    direct = StableVideoUNet(unet=SVDUNetHIP(cfg, sd, DEV), timesteps=StableVideoUNet._default_timestep_schedule(25))
    assert len(loaded.timesteps) == 25 and loaded.unet.cfg == cfg
    g = torch.Generator().manual_seed(1)
    emb = torch.randn(1, 1, cfg.cross_attention_dim, generator=g).half().to(DEV)
    img = torch.randn(1, 4, 3, 8, 8, generator=g).half().to(DEV)
    lat = torch.randn(1, 4, 3, 8, 8, generator=g).half().to(DEV)
    for m in (loaded, direct):
        m.set_conditioning(emb, img, num_frames=3)
    assert torch.equal(loaded(lat, 5), direct(lat, 5))

#!/usr/bin/env python3
"""Per-video forward time: batch-B latents vs B interleaved streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd.models.svd_unet import StableVideoUNet
dev = torch.device("cuda:0")
model = StableVideoUNet.from_random_init(StableVideoUNet._default_timestep_schedule(25), seed=0, device=dev)
steps = 6
ALL = [torch.cuda.Stream() for _ in range(4)]
for B, S in [tuple(map(int, a.split("x"))) for a in os.environ.get("ARMS", "1x1,1x2,1x3,2x1,2x2,1x2,1x1").split(",")]:
    torch.manual_seed(42)
    model.set_dummy_conditioning(B, 14, 72, 128, dev)
    lats = [torch.randn(B, 4, 14, 72, 128, device=dev, dtype=torch.float16) * model.init_noise_sigma for _ in range(S)]
    streams = ALL[:S] if os.environ.get('FRESH') != '1' else [torch.cuda.Stream() for _ in range(S)]
    def run():
        xs = list(lats)
        for s in range(steps):
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    xs[i] = model(xs[i], s)
        for st in streams: st.synchronize()
    with torch.no_grad():
        run(); torch.cuda.synchronize()
        t = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"batch {B} x {S} streams: {dt*1e3/(steps*B*S):7.2f} ms per video-forward", flush=True)

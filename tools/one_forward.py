#!/usr/bin/env python3
"""Two SVD UNet steps (1 warm-up + 1 profiled) at the benchmark shape and micro-batch (BATCH, default 2 = bench.py's
default); used under rocprofv3 --pmc."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd.models.svd_unet import StableVideoUNet
dev = torch.device("cuda:0")
frames = int(os.environ.get("FRAMES", 14))
batch = int(os.environ.get("BATCH", 2))
model = StableVideoUNet.from_random_init(StableVideoUNet._default_timestep_schedule(25), seed=0, device=dev)
torch.manual_seed(42)
model.set_dummy_conditioning(batch, frames, 72, 128, dev)
lat = torch.randn(batch, 4, frames, 72, 128, device=dev, dtype=torch.float16) * model.init_noise_sigma
with torch.no_grad():
    for step in range(int(os.environ.get("NSTEPS", 2))):
        lat = model(lat, step)
torch.cuda.synchronize()
print("done", float(lat.float().abs().mean()))

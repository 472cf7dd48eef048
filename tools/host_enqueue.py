#!/usr/bin/env python3
"""Host time to ENQUEUE one SVD UNet step (no synchronisation) vs the GPU time it takes: shows how far the Python
launcher runs ahead of the device (if enqueue >= GPU time / lanes, HIP-graph replay would pay)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd.models.svd_unet import StableVideoUNet
dev = torch.device("cuda:0")
model = StableVideoUNet.from_random_init(StableVideoUNet._default_timestep_schedule(25), seed=0, device=dev)
torch.manual_seed(42)
model.set_dummy_conditioning(1, 14, 72, 128, dev)
lat = torch.randn(1, 4, 14, 72, 128, device=dev, dtype=torch.float16) * model.init_noise_sigma
with torch.no_grad():
    for s in range(3): lat = model(lat, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(3, 13): lat = model(lat, s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / 10:.1f} ms per step, GPU {1e3 * (t2 - t0) / 10:.1f} ms per step")

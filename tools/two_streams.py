#!/usr/bin/env python3
"""Does interleaving two videos on two HIP streams raise throughput on one GPU?"""
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
lats = [torch.randn(1, 4, 14, 72, 128, device=dev, dtype=torch.float16) * model.init_noise_sigma for _ in range(2)]
steps = 6
def seq():
    for l in lats:
        x = l
        for s in range(steps): x = model(x, s)
STREAMS = [torch.cuda.Stream() for _ in range(3)]
def par(n=2):
    streams = STREAMS[:n]
    xs = [lats[i % 2] for i in range(n)]
    for s in range(steps):
        for i in range(n):
            with torch.cuda.stream(streams[i]):
                xs[i] = model(xs[i], s)
    for st in streams: st.synchronize()
with torch.no_grad():
    for rep in range(3):
        for fn, name, nv in ((seq, "sequential", 2), (lambda: par(2), "two streams", 2), (lambda: par(3), "three streams", 3)):
            torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
            dt = time.perf_counter() - t
            print(f"{name:14s}: {dt*1e3/(nv*steps):7.2f} ms per forward", flush=True)

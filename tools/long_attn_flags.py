#!/usr/bin/env python3
"""How often the frozen-reference attention kernel's speculation fails inside the UNet: runs NSTEPS denoising steps at the
benchmark shape (random-init weights, as bench.py) and counts, per attention call, the 256-row blocks that were flagged and
recomputed by the ordinary kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd.hip import ops
from vdpp_amd.models.svd_unet import StableVideoUNet
dev = torch.device("cuda:0")
batch = int(os.environ.get("BATCH", 2))
model = StableVideoUNet.from_random_init(StableVideoUNet._default_timestep_schedule(25), seed=0, device=dev, long_attention=True)
torch.manual_seed(42)
model.set_dummy_conditioning(batch, 14, 72, 128, dev)
lat = torch.randn(batch, 4, 14, 72, 128, device=dev, dtype=torch.float16) * model.init_noise_sigma
calls = []
orig = ops.attn_spatial_long
def counted(q, k, v, o, ws, **kw):
    r = orig(q, k, v, o, ws, **kw)
    torch.cuda.synchronize()
    n = ops.attn_long_ws_bytes(kw["batch"], kw["seq"], kw["heads"]) // 4 - 1      # (the last word counts flagged waves)
    calls.append((kw["batch"], kw["seq"], kw["heads"], int(ws.view(torch.int32)[:n].sum()), n))
    return r
ops.attn_spatial_long = counted
with torch.no_grad():
    for step in (0, 12, 24)[:int(os.environ.get("NSTEPS", 3))]:
        calls.clear()
        out = model(lat, step)
        print(f"step {step}: " + ", ".join(f"{b}x{s}x{h}: {f}/{n}" for b, s, h, f, n in calls), flush=True)
torch.cuda.synchronize()

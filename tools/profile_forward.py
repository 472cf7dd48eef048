#!/usr/bin/env python3
"""Per-shape timing of one SVD UNet forward (events around every contraction/attention launch)."""
import os, sys, json, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd.hip import ops
from vdpp_amd.models.svd_unet import StableVideoUNet

def main():
    dev = torch.device("cuda:0")
    frames = int(os.environ.get("FRAMES", 14))
    model = StableVideoUNet.from_random_init(StableVideoUNet._default_timestep_schedule(25), seed=0, device=dev)
    model.set_dummy_conditioning(1, frames, 72, 128, dev)
    lat = torch.randn(1, 4, frames, 72, 128, device=dev, dtype=torch.float16) * model.init_noise_sigma
    with torch.no_grad():
        model(lat, 0); model(lat, 1)
        torch.cuda.synchronize()
        ops.PROFILE = []
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); model(lat, 2); t1.record()
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
    print("forward ms (with event overhead):", t0.elapsed_time(t1))
    agg = collections.OrderedDict()
    for kind, fl, e0, e1, _nb, tag in prof:
        ms = e0.elapsed_time(e1)
        if kind == "gemm":
            key = ("gemm",) + tag
        else:
            key = (kind, fl or _nb)
        a = agg.setdefault(key, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += fl; a[3] += _nb
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for v in agg.values())
    print(f"total timed {tot:.2f} ms")
    for k, (n, ms, fl, nb) in rows[:80]:
        print(f"{str(k):60s} x{n:3d} {ms:8.3f} ms  {ms/n*1000:9.1f} us/launch  {fl/ms/1e9:8.1f} TF/s  {nb/ms/1e9:7.2f} TB/s (algorithmic)")
    kinds = collections.OrderedDict()
    for k, (n, ms, fl, nb) in agg.items():
        a = kinds.setdefault(k[0], [0, 0.0]); a[0] += n; a[1] += ms
    for k, (n, ms) in sorted(kinds.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:16s} {n:4d} launches {ms:8.3f} ms")

if __name__ == "__main__":
    main()

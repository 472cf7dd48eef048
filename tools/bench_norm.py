#!/usr/bin/env python3
"""Micro-benchmark of GroupNorm / LayerNorm kernels at SVD shapes; prints GB/s (algorithmic bytes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd.hip import ops
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it
for inst, rows, c in [(14, 9216, 320), (1, 129024, 320), (14, 2304, 640), (14, 2304, 1920), (14, 576, 1280), (14, 9216, 640), (14, 9216, 960), (14, 144, 1280), (14, 576, 2560), (14, 144, 2560), (14, 2304, 1280)]:
    x = torch.randn(inst * rows, c, device="cuda", dtype=torch.float16)
    y = torch.empty_like(x)
    g = torch.ones(c, device="cuda"); b = torch.zeros(c, device="cuda")
    ws = torch.empty(ops.groupnorm_ws_bytes(inst, rows, c, 32), dtype=torch.uint8, device="cuda")
    us = timeit(lambda: ops.groupnorm(x, g, b, y, instances=inst, rows=rows, c=c, groups=32, eps=1e-5, silu=True, ws=ws))
    print(f"groupnorm {inst:3d}x{rows:6d}x{c:4d}: {us:8.1f} us  {3*x.numel()*2/us/1e3:8.1f} GB/s (2 reads + 1 write)", flush=True)
for rows, c in [(129024, 320), (32256, 640), (8064, 1280)]:
    x = torch.randn(rows, c, device="cuda", dtype=torch.float16); y = torch.empty_like(x)
    g = torch.ones(c, device="cuda"); b = torch.zeros(c, device="cuda")
    us = timeit(lambda: ops.layernorm(x, g, b, y, rows=rows, c=c))
    print(f"layernorm {rows:6d}x{c:4d}: {us:8.1f} us  {2*x.numel()*2/us/1e3:8.1f} GB/s", flush=True)

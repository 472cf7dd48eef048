#!/usr/bin/env python3
"""Micro-benchmark of sp_gemm_f16 on chosen shapes: MODE M N CIN [geglu] per argument 'mode:m:n:cin[:g]'."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd.hip import ops

def run(spec, iters=20):
    parts = spec.split(":")
    mode, m, n, cin = int(parts[0]), int(parts[1]), int(parts[2]), int(parts[3])
    geglu = len(parts) > 4 and parts[4] == "g"
    dev = "cuda"
    taps = {0: 1, 1: 9, 2: 3}[mode]
    conv = temporal = None
    if mode == 1:
        h, w = 72, 128
        while 14 * h * w > m: h //= 2; w //= 2
        assert 14 * h * w == m, (h, w, m)
        conv = (14, h, w, h, w, 1, 0)
    if mode == 2:
        temporal = (14, m // 14)
    a = torch.randn(m, cin, device=dev, dtype=torch.float16)
    wt = torch.randn(n, taps * cin, device=dev, dtype=torch.float16) * 0.02
    out = torch.empty(m, n // 2 if geglu else n, device=dev, dtype=torch.float16)
    bias = torch.randn(n, device=dev)
    kw = dict(m=m, n=n, cin=cin, mode=mode, conv=conv, temporal=temporal, bias=bias, geglu=geglu)
    for _ in range(3): ops.gemm(a, wt, out, **kw)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm(a, wt, out, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * m * n * taps * cin
    print(f"{spec:28s} {us:9.1f} us  {fl/us/1e6:8.1f} TF/s", flush=True)

if __name__ == "__main__":
    iters = int(os.environ.get("ITERS", 20))
    for s in sys.argv[1:]:
        run(s, iters)

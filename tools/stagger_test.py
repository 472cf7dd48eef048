#!/usr/bin/env python3
"""Lock-step experiment on the ping-pong GEMM (needs `make -C csrc exp`): does starting the first round of
workgroups in four phase groups (SP_GEMM_STAGGER = microseconds between groups) shorten short-K GEMMs?
Interleaved rounds in one process.   usage: stagger_test.py mode:m:n:cin[:g] ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd import hip
hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), "libsvdpipe_hip_exp.so")
from vdpp_amd.hip import ops

STAGGERS = [int(x) for x in os.environ.get("STAGGERS", "0,2,4,6,9").split(",")]


def run(spec, iters=20, rounds=5):
    parts = spec.split(":")
    mode, m, n, cin = int(parts[0]), int(parts[1]), int(parts[2]), int(parts[3])
    geglu = "g" in parts[4:]
    dev = "cuda"
    taps = {0: 1, 1: 9, 2: 3}[mode]
    conv = temporal = None
    if mode == 1:
        h, w = 72, 128
        while 14 * h * w > m: h //= 2; w //= 2
        conv = (14, h, w, h, w, 1, 0)
    if mode == 2:
        temporal = (14, m // 14)
    a = torch.randn(m, cin, device=dev, dtype=torch.float16)
    wt = torch.randn(n, taps * cin, device=dev, dtype=torch.float16) * 0.02
    out = torch.empty(m, n // 2 if geglu else n, device=dev, dtype=torch.float16)
    bias = torch.randn(n, device=dev)
    kw = dict(m=m, n=n, cin=cin, mode=mode, conv=conv, temporal=temporal, bias=bias, geglu=geglu)
    best = {s: 1e9 for s in STAGGERS}
    for r in range(rounds):
        for s in STAGGERS:
            os.environ["SP_GEMM_STAGGER"] = str(s)
            for _ in range(2): ops.gemm(a, wt, out, **kw)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): ops.gemm(a, wt, out, **kw)
            e1.record(); torch.cuda.synchronize()
            best[s] = min(best[s], e0.elapsed_time(e1) * 1e3 / iters)
    fl = 2.0 * m * n * taps * cin
    print(f"{spec:28s} " + "  ".join(f"s={s}: {best[s]:7.1f} us {fl/best[s]/1e6:6.0f} TF" for s in STAGGERS), flush=True)


if __name__ == "__main__":
    for s in sys.argv[1:]:
        run(s)

#!/usr/bin/env python3
"""Timing-only experiment (needs `make -C csrc exp`): how much faster would the conv3x3 / temporal-conv GEMMs run if
the A operand of taps 1..8 (1..2) did not have to be fetched again through the CU's memory pipe (an LDS-resident halo
tile)?  SP_GEMM_DBG=8 makes those taps read the zero page (same instruction count, one cache line instead of 16)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd import hip
hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), "libsvdpipe_hip_exp.so")
from vdpp_amd.hip import ops


def run(spec, iters=10, rounds=3):
    parts = spec.split(":")
    mode, m, n, cin = int(parts[0]), int(parts[1]), int(parts[2]), int(parts[3])
    taps = {0: 1, 1: 9, 2: 3}[mode]
    conv = temporal = None
    if mode == 1:
        h, w = 72, 128
        while 14 * h * w > m: h //= 2; w //= 2
        conv = (14, h, w, h, w, 1, 0)
    if mode == 2:
        temporal = (14, m // 14)
    a = torch.randn(m, cin, device="cuda", dtype=torch.float16)
    wt = torch.randn(n, taps * cin, device="cuda", dtype=torch.float16) * 0.02
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    kw = dict(m=m, n=n, cin=cin, mode=mode, conv=conv, temporal=temporal, bias=torch.randn(n, device="cuda"))
    best = {0: 1e9, 8: 1e9}
    with ops.gemm_route(2):
        for r in range(rounds):
            for dbg in (0, 8):
                os.environ["SP_GEMM_DBG"] = str(dbg)
                for _ in range(2): ops.gemm(a, wt, out, **kw)
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters): ops.gemm(a, wt, out, **kw)
                e1.record(); torch.cuda.synchronize()
                best[dbg] = min(best[dbg], e0.elapsed_time(e1) * 1e3 / iters)
    os.environ["SP_GEMM_DBG"] = "0"
    fl = 2.0 * m * n * taps * cin
    print(f"{spec:26s} real {best[0]:8.1f} us {fl/best[0]/1e6:6.0f} TF   taps>0 from zero page {best[8]:8.1f} us {fl/best[8]/1e6:6.0f} TF  x{best[0]/best[8]:.2f}", flush=True)


if __name__ == "__main__":
    for s in sys.argv[1:]:
        run(s)

#!/usr/bin/env python3
"""Persistent-stream GEMM with the workgroups' start times spread over G phase groups D us apart (needs `make -C csrc
exp`): does taking the CUs out of lock-step (no chip-wide store bursts) shorten the kernel?  Best of interleaved rounds.
usage: ps_stagger.py mode:m:n:cin[:g][:r] ...   env BM, ARMS="0,4x3,8x2" (GxD)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd import hip
hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), "libsvdpipe_hip_exp.so")
from vdpp_amd.hip import ops

ARMS = os.environ.get("ARMS", "0,4x2,4x3,4x4,8x1,8x2,16x1").split(",")


def code(arm):
    if arm == "0": return 0
    g, d = arm.split("x")
    return int(g) * 256 + int(d)


def run(spec, iters=20, rounds=4):
    parts = spec.split(":")
    mode, m, n, cin = int(parts[0]), int(parts[1]), int(parts[2]), int(parts[3])
    flags = parts[4:]
    geglu = "g" in flags
    dev = "cuda"
    a = torch.randn(m, cin, device=dev, dtype=torch.float16)
    wt = torch.randn(n, cin, device=dev, dtype=torch.float16) * 0.02
    no = n // 2 if geglu else n
    out = torch.empty(m, no, device=dev, dtype=torch.float16)
    kw = dict(m=m, n=n, cin=cin, mode=mode, bias=torch.randn(n, device=dev), geglu=geglu)
    if "r" in flags:
        kw.update(res1=torch.randn(m, no, device=dev, dtype=torch.float16), r1scale=1.0)
    best = {arm: 1e9 for arm in ARMS}
    bm = int(os.environ.get("BM", 256))
    with ops.gemm_route(3, bm=bm):
        for r in range(rounds):
            for arm in ARMS:
                os.environ["SP_GEMM_STAGGER"] = str(code(arm))
                for _ in range(2): ops.gemm(a, wt, out, **kw)
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters): ops.gemm(a, wt, out, **kw)
                e1.record(); torch.cuda.synchronize()
                best[arm] = min(best[arm], e0.elapsed_time(e1) * 1e3 / iters)
    os.environ["SP_GEMM_STAGGER"] = "0"
    print(f"{spec:28s} " + "  ".join(f"{arm}: {best[arm]:6.1f}" for arm in ARMS), flush=True)


if __name__ == "__main__":
    for s in sys.argv[1:]:
        run(s)

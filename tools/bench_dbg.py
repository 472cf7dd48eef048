#!/usr/bin/env python3
"""A/B of SP_GEMM_DBG ablation bits on the experiments library (`make -C csrc exp`), interleaved rounds in one process.
usage: bench_dbg.py mode:m:n:cin[:g][:r] ...   env ROUTE=3 BM=256 ARMS="0,16"  (16: K-steps staged one by one)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd import hip
hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), "libsvdpipe_hip_exp.so")
from vdpp_amd.hip import ops

ARMS = os.environ.get("ARMS", "0,16").split(",")


def run(spec, iters=20, rounds=5):
    parts = spec.split(":")
    mode, m, n, cin = int(parts[0]), int(parts[1]), int(parts[2]), int(parts[3])
    flags = parts[4:]
    geglu = "g" in flags
    dev = "cuda"
    taps = {0: 1, 1: 9, 2: 3}[mode]
    conv = temporal = None
    if mode == 1:
        h, w = 72, 128
        nimg = 14
        while nimg * h * w > m and h > 9: h //= 2; w //= 2
        while nimg * h * w < m: nimg += 14                # micro-batches: more images of the smallest level
        conv = (nimg, h, w, h, w, 1, 0)
    if mode == 2:
        temporal = (14, m // 14) if m <= 129024 else (14, 129024 // 14)
    a = torch.randn(m, cin, device=dev, dtype=torch.float16)
    wt = torch.randn(n, taps * cin, device=dev, dtype=torch.float16) * 0.02
    no = n // 2 if geglu else n
    out = torch.empty(m, no, device=dev, dtype=torch.float16)
    kw = dict(m=m, n=n, cin=cin, mode=mode, conv=conv, temporal=temporal, bias=torch.randn(n, device=dev), geglu=geglu)
    if "r" in flags:
        kw.update(res1=torch.randn(m, no, device=dev, dtype=torch.float16), r1scale=1.0)
    best = {arm: 1e9 for arm in ARMS}
    ref = None
    with ops.gemm_route(int(os.environ.get("ROUTE", 3)), bm=int(os.environ.get("BM", 0))):
        for r in range(rounds):
            for arm in ARMS:
                os.environ["SP_GEMM_DBG"] = arm
                for _ in range(2): ops.gemm(a, wt, out, **kw)
                torch.cuda.synchronize()
                if r == 0:
                    if ref is None: ref = out.float().clone()
                    elif not os.environ.get("NOCHECK"): assert torch.equal(out.float(), ref), (spec, arm, float((out.float() - ref).abs().max()))
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters): ops.gemm(a, wt, out, **kw)
                e1.record(); torch.cuda.synchronize()
                best[arm] = min(best[arm], e0.elapsed_time(e1) * 1e3 / iters)
    fl = 2.0 * m * n * taps * cin
    print(f"{spec:28s} " + "  ".join(f"dbg={arm}: {best[arm]:7.1f} us {fl / best[arm] / 1e6:5.0f} TF" for arm in ARMS), flush=True)


if __name__ == "__main__":
    for s in sys.argv[1:]:
        run(s)

#!/usr/bin/env python3
"""Shader clock the chip holds under each KIND of kernel of the UNet step, measured from the outside: every arm launches one
kernel back to back for SECONDS (default 1.0) on random data between two `sp_clock_stamp`s (s_memtime against the 100 MHz
counter, per XCD), and prints the clock, the launch time and the rate.  Shapes are those of a micro-batch of two videos.
usage: clock_by_kernel.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd.hip import ops

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device("cuda:0")


def gemm_arm(mode, m, n, cin, geglu=False, res=False):
    taps = {0: 1, 1: 9}[mode]
    conv = None
    if mode == 1:
        h, w, nimg = 72, 128, 28
        while nimg * h * w > m: h //= 2; w //= 2
        conv = (nimg, h, w, h, w, 1, 0)
    a = torch.randn(m, cin, device=dev, dtype=torch.float16)
    wt = torch.randn(n, taps * cin, device=dev, dtype=torch.float16) * 0.02
    no = n // 2 if geglu else n
    out = torch.empty(m, no, device=dev, dtype=torch.float16)
    kw = dict(m=m, n=n, cin=cin, mode=mode, conv=conv, bias=torch.randn(n, device=dev), geglu=geglu)
    if res:
        kw.update(res1=torch.randn(m, no, device=dev, dtype=torch.float16), r1scale=1.0)
    return (lambda: ops.gemm(a, wt, out, **kw)), 2.0 * m * n * taps * cin, 2.0 * (m * cin + m * no * (2 if res else 1))


def attn_arm(batch, seq, heads):
    c = heads * 64
    qkv = torch.randn(batch * seq, 3 * c, device=dev, dtype=torch.float16)
    o = torch.empty(batch * seq, c, device=dev, dtype=torch.float16)
    ws = torch.zeros(ops.attn_long_ws_bytes(batch, seq, heads) // 4 + 1, dtype=torch.int32, device=dev)
    f = lambda: ops.attn_spatial_long(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], o, ws, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c,
                                      batch=batch, seq=seq, heads=heads)
    return f, 4.0 * batch * heads * seq * seq * 64, 2.0 * 4 * batch * seq * c


def gn_arm(instances, rows, c):
    x = torch.randn(instances * rows, c, device=dev, dtype=torch.float16)
    y = torch.empty_like(x)
    g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    ws = torch.empty(ops.groupnorm_ws_bytes(instances, rows, c, 32), dtype=torch.uint8, device=dev)
    f = lambda: ops.groupnorm(x, g, b, y, instances=instances, rows=rows, c=c, groups=32, eps=1e-5, silu=True, ws=ws)
    return f, 0.0, 3 * 2.0 * instances * rows * c


def copy_arm(n):
    x = torch.randn(n, device=dev, dtype=torch.float16)
    y = torch.empty_like(x)
    return (lambda: y.copy_(x)), 0.0, 4.0 * n


ARMS = [
    ("conv3x3 level 0  258048 x 320 x 2880", lambda: gemm_arm(1, 258048, 320, 320)),
    ("conv3x3 level 2   16128 x 1280 x 11520", lambda: gemm_arm(1, 16128, 1280, 1280)),
    ("FF1 GEGLU level 0  258048 x 2560 x 320", lambda: gemm_arm(0, 258048, 2560, 320, geglu=True)),
    ("FF1 GEGLU level 2  16128 x 10240 x 1280", lambda: gemm_arm(0, 16128, 10240, 1280, geglu=True)),
    ("linear + residual level 0  258048 x 320 x 320", lambda: gemm_arm(0, 258048, 320, 320, res=True)),
    ("spatial attention level 0  28 x 9216 x 5 heads", lambda: attn_arm(28, 9216, 5)),
    ("GroupNorm + SiLU level 0  28 x 9216 x 320", lambda: gn_arm(28, 9216, 320)),
    ("device copy 165 MB (torch)", lambda: copy_arm(258048 * 320)),
]

print(f"{'kernel':50s} {'us/launch':>10s} {'TFLOP/s':>8s} {'TB/s alg.':>9s} {'GHz':>6s} {'XCDs':>4s}")
for name, make in ARMS:
    f, flop, byts = make()
    for _ in range(5): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); torch.cuda.synchronize()
    one = max(time.perf_counter() - t0, 2e-5)
    reps = max(20, int(SECONDS / one))
    for _ in range(reps // 4): f()                      # settle the clock under this load before the stamps
    st = ops.ClockStamps(dev, 2)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    st.stamp(); e0.record()
    for _ in range(reps): f()
    e1.record(); st.stamp()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    ghz, secs, xcds = st.ghz()
    print(f"{name:50s} {us:10.1f} {flop / us / 1e6:8.0f} {byts / us / 1e6:9.2f} {ghz:6.3f} {xcds:4d}", flush=True)
    del f
    torch.cuda.empty_cache()

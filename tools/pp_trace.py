#!/usr/bin/env python3
"""Phase breakdown of the ping-pong GEMM per workgroup (needs `make -C csrc exp`: loads libsvdpipe_hip_exp.so).

Each workgroup records a 100 MHz wall-clock stamp at entry, after the prologue, after the K loop and after the
epilogue, plus its hardware id; this prints the mean of each phase and the idle gap between consecutive
workgroups on the same CU.   usage: pp_trace.py mode:m:n:cin[:g][:r] ...   (r = with residual)
"""
import os, sys, ctypes, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import vdpp_amd  # noqa
from vdpp_amd import hip
hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), "libsvdpipe_hip_exp.so")
from vdpp_amd.hip import ops


def run(spec):
    parts = spec.split(":")
    mode, m, n, cin = int(parts[0]), int(parts[1]), int(parts[2]), int(parts[3])
    flags = parts[4:]
    geglu, resid = "g" in flags, "r" in flags
    dev = "cuda"
    taps = {0: 1, 1: 9, 2: 3}[mode]
    nk = taps * cin // 32
    conv = temporal = None
    if mode == 1:
        h, w = 72, 128
        while 14 * h * w > m: h //= 2; w //= 2
        conv = (14, h, w, h, w, 1, 0)
    if mode == 2:
        temporal = (14, m // 14)
    a = torch.randn(m, cin, device=dev, dtype=torch.float16)
    wt = torch.randn(n, taps * cin, device=dev, dtype=torch.float16) * 0.02
    no = n // 2 if geglu else n
    out = torch.empty(m, no, device=dev, dtype=torch.float16)
    bias = torch.randn(n, device=dev)
    kw = dict(m=m, n=n, cin=cin, mode=mode, conv=conv, temporal=temporal, bias=bias, geglu=geglu)
    if resid:
        kw["res1"] = torch.randn(m, no, device=dev, dtype=torch.float16)
    bm = int(os.environ.get("BM", 256))
    hip.load().sp_gemm_set_route(2, bm, 0)          # the stamps live in gemm_pp.hip
    for _ in range(3): ops.gemm(a, wt, out, **kw)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm(a, wt, out, **kw); e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    lib = hip.load()
    nw = 16384
    buf = np.zeros((nw, 12), dtype=np.int64)
    rc = lib.sp_debug_pp_trace(buf.ctypes.data_as(ctypes.c_void_p), nw)
    assert rc == 0, rc
    used = buf[:, 0] > 0
    # only the workgroups of this launch: stamps within the last launch window
    tmax = buf[used, 3].max()
    sel = used & (buf[:, 0] > tmax - int(us * 100 * 1.5))
    t = buf[sel].astype(np.float64)
    t0 = t[:, 0].min()
    pro, main, epi = (t[:, 1] - t[:, 0]) / 100, (t[:, 2] - t[:, 1]) / 100, (t[:, 3] - t[:, 2]) / 100
    span = (t[:, 3].max() - t0) / 100
    cu = (buf[sel, 5] >> 32) * 65536 + (buf[sel, 5] & 0xFF00 & 0xFFFF) + ((buf[sel, 5] >> 13) & 7) * 16
    gaps = []
    per_cu = collections.defaultdict(list)
    for i, c in enumerate(cu): per_cu[int(c)].append((t[i, 0], t[i, 3]))
    for c, lst in per_cu.items():
        lst.sort()
        for (s0, e0_), (s1, e1_) in zip(lst, lst[1:]): gaps.append((s1 - e0_) / 100)
    first = (t[:, 0].min() - t0) / 100
    print(f"{spec:30s} {us:8.1f} us  wgs {sel.sum():5d} cus {len(per_cu):3d}  span {span:7.1f}  "
          f"prologue {pro.mean():5.2f}  loop {main.mean():6.2f}  epilogue {epi.mean():5.2f}  "
          f"gap {np.mean(gaps) if gaps else 0:5.2f} (p90 {np.percentile(gaps, 90) if gaps else 0:5.2f})  "
          f"start-spread {(t[:, 0].max() - t0) / 100:6.1f}\n"
          f"{'':30s} prologue parts: setup {np.mean(t[:, 10] - t[:, 0]) / 100:5.2f}  dma-issue {np.mean(t[:, 11] - t[:, 10]) / 100:5.2f}  "
          f"acc-init+wait+barrier {np.mean(t[:, 1] - t[:, 11]) / 100:5.2f}\n"
          f"{'':30s} K loop: {np.mean(t[:, 9] - t[:, 8]) / max(1, nk):8.0f} shader cycles per K-step, "
          f"clock {np.mean((t[:, 9] - t[:, 8]) / np.maximum(t[:, 2] - t[:, 1], 1)) * 100:6.0f} MHz\n"
          f"{'':30s} epilogue parts: res-issue {np.mean(t[:, 4] - t[:, 2]) / 100:5.2f}  stage+barrier {np.mean(t[:, 6] - t[:, 4]) / 100:5.2f}  "
          f"lds-read+residual {np.mean(t[:, 7] - t[:, 6]) / 100:5.2f}  stores {np.mean(t[:, 3] - t[:, 7]) / 100:5.2f}", flush=True)


if __name__ == "__main__":
    for s in sys.argv[1:]:
        run(s)

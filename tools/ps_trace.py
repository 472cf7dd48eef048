#!/usr/bin/env python3
"""Where a persistent-stream GEMM workgroup spends its time (needs `make -C csrc exp`): per tile, the K loop, the
re-alignment barriers of the two wave halves and the epilogue, for waves 0 (early half) and 4 (late half).
usage: ps_trace.py mode:m:n:cin[:g][:r][:r2] ...   env BM=256|192"""
import os, sys, ctypes
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
    geglu = "g" in flags
    dev = "cuda"
    a = torch.randn(m, cin, device=dev, dtype=torch.float16)
    wt = torch.randn(n, cin, device=dev, dtype=torch.float16) * 0.02
    no = n // 2 if geglu else n
    out = torch.empty(m, no, device=dev, dtype=torch.float16)
    kw = dict(m=m, n=n, cin=cin, mode=mode, bias=torch.randn(n, device=dev), geglu=geglu)
    if "r" in flags or "r2" in flags:
        kw.update(res1=torch.randn(m, no, device=dev, dtype=torch.float16), r1scale=1.0)
    if "r2" in flags:
        kw.update(res2=torch.randn(m, no, device=dev, dtype=torch.float16), r2scale=0.5)
    bm = int(os.environ.get("BM", 256))
    with ops.gemm_route(3, bm=bm):
        for _ in range(3): ops.gemm(a, wt, out, **kw)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm(a, wt, out, **kw); e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    buf = np.zeros((256, 16), dtype=np.int64)
    rc = hip.load().sp_debug_ps_trace(buf.ctypes.data_as(ctypes.c_void_p), 256)
    assert rc == 0, rc
    t = buf.astype(np.float64) / 100.0          # us
    for half, o in (("waves 0-3", 0), ("waves 4-7", 8)):
        tiles = t[:, o + 6] * 100
        sel = tiles > 0
        nt = tiles[sel]
        print(f"{spec:28s} {us:8.1f} us  {half}: tiles/wg {nt.mean():5.2f}  start->first barrier {np.mean(t[sel, o + 1] - t[sel, o]):5.2f}  "
              f"per tile: K loop {np.mean(t[sel, o + 2] / nt):6.2f}  realign {np.mean(t[sel, o + 3] / nt):5.2f}  "
              f"epilogue {np.mean(t[sel, o + 4] / nt):5.2f}   wg life {np.mean(t[sel, o + 5] - t[sel, o]):7.1f} "
              f"(max {np.max(t[sel, o + 5] - t[sel, o]):7.1f})", flush=True)


    if os.environ.get("STEPS"):
        nk = cin // 32
        sb = np.zeros((256, 2, 64), dtype=np.int64)
        rc = hip.load().sp_debug_ps_steps(sb.ctypes.data_as(ctypes.c_void_p), 256)
        assert rc == 0, rc
        st = sb.astype(np.float64) / 100.0
        ok = sb[:, 0, 0] > 0
        for half in (0, 1):
            d = np.diff(st[ok, half, :nk + 3], axis=1)         # nk K-steps, realign, epilogue
            print("   second tile, waves %s: K-steps " % ("0-3" if half == 0 else "4-7") + " ".join(f"{x:5.2f}" for x in d[:, :nk].mean(0))
                  + f" | realign {d[:, nk].mean():5.2f} | epilogue {d[:, nk + 1].mean():5.2f}", flush=True)


if __name__ == "__main__":
    for s in sys.argv[1:]:
        run(s)

#!/usr/bin/env python3
"""A/B of the spatial attention kernel against the previous build of it (linked into libsvdpipe_hip_exp.so as
sp_attn_spatial_f16_old), interleaved rounds in one process.  args 'batch:seq:heads'."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd import hip
hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), "libsvdpipe_hip_exp.so")
from vdpp_amd.hip import ops
lib = hip.load()
old = lib.sp_attn_spatial_f16_old
old.restype = ctypes.c_int
old.argtypes = hip.SIGNATURES["sp_attn_spatial_f16"][1]
for spec in sys.argv[1:]:
    b, s, h = map(int, spec.split(":"))
    c = h * 64
    qkv = torch.randn(b * s, 3 * c, device="cuda", dtype=torch.float16)
    o = torch.empty(b * s, c, device="cuda", dtype=torch.float16)
    o2 = torch.empty_like(o)
    zp = ops.zero_page(o.device).data_ptr()
    q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
    st = torch.cuda.current_stream().cuda_stream
    def run_new(): ops.attn_spatial(q, k, v, o, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=b, seq=s, heads=h)
    def run_old(): assert old(q.data_ptr(), k.data_ptr(), v.data_ptr(), o2.data_ptr(), 3 * c, 3 * c, 3 * c, c, b, s, h, 0.125, zp, st) == 0
    best = {"new": 1e9, "old": 1e9}
    for r in range(4):
        for name, fn in (("old", run_old), ("new", run_new)):
            for _ in range(2): fn()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            best[name] = min(best[name], e0.elapsed_time(e1) * 1e3 / 10)
    err = float((o.float() - o2.float()).norm() / o2.float().norm())
    fl = 4.0 * b * h * s * s * 64
    print(f"{spec:16s} old {best['old']:9.1f} us {fl/best['old']/1e6:7.1f} TF/s   new {best['new']:9.1f} us {fl/best['new']/1e6:7.1f} TF/s  x{best['old']/best['new']:.3f}  rel diff {err:.2e}", flush=True)

#!/usr/bin/env python3
"""Micro-benchmark of sp_attn_spatial_f16 and sp_attn_spatial_fp8 (incl. its quantise pass): args 'batch:seq:heads'."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd.hip import ops
for spec in sys.argv[1:]:
    b, s, h = map(int, spec.split(":"))
    c = h * 64
    qkv = torch.randn(b * s, 3 * c, device="cuda", dtype=torch.float16)
    o = torch.empty(b * s, c, device="cuda", dtype=torch.float16)
    kw = dict(ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=b, seq=s, heads=h)
    for _ in range(3): ops.attn_spatial(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], o, **kw)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    it = 10
    e0.record()
    for _ in range(it): ops.attn_spatial(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], o, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / it
    ws = torch.empty(ops.attn_fp8_ws_bytes(b, s, h), dtype=torch.uint8, device="cuda")
    for _ in range(3): ops.attn_spatial_fp8(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], o, ws, **kw)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(it): ops.attn_spatial_fp8(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], o, ws, **kw)
    e1.record(); torch.cuda.synchronize()
    us8 = e0.elapsed_time(e1) * 1e3 / it
    print(f"{spec:16s} f16 {us:9.1f} us {4.0*b*h*s*s*64/us/1e6:8.1f} TF/s   fp8 {us8:9.1f} us {4.0*b*h*s*s*64/us8/1e6:8.1f} TF/s",
          flush=True)

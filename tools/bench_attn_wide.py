#!/usr/bin/env python3
"""The one-wave-per-SIMD attention experiment (tools/experiments/attention_wide.hip) against the product kernel, one
process.  Build first:  cd video-diffusion-pipeline-parallel_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950
-I. -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -c ../../tools/experiments/attention_wide.hip -o
build_exp/attn_wide_old.o && make exp.   usage: bench_attn_wide.py batch:seq:heads  [qb:variant ...]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd import hip
hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), "libsvdpipe_hip_exp.so")
from vdpp_amd.hip import ops
lib = hip.load()
wide = lib.sp_exp_attn_wide
wide.restype = ctypes.c_int
P, L, I, F = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
wide.argtypes = [P, P, P, P, L, L, L, L, I, I, I, F, I, I, P]
b, s, h = map(int, sys.argv[1].split(":"))
arms = [tuple(map(int, a.split(":"))) for a in sys.argv[2:]] or [(3, 0), (4, 0)]
c = h * 64
qkv = torch.randn(b * s, 3 * c, device="cuda", dtype=torch.float16)
q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
o = torch.empty(b * s, c, device="cuda", dtype=torch.float16)
o2 = torch.empty_like(o)
st = torch.cuda.current_stream().cuda_stream
def base(): ops.attn_spatial(q, k, v, o, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=b, seq=s, heads=h)
def mk(qb, var):
    def f():
        rc = wide(q.data_ptr(), k.data_ptr(), v.data_ptr(), o2.data_ptr(), 3 * c, 3 * c, 3 * c, c, b, s, h, 0.125, qb, var, st)
        assert rc == 0, rc
    return f
ws = torch.empty(ops.attn_long_ws_bytes(b, s, h), dtype=torch.uint8, device="cuda")
def bounded(): ops.attn_spatial_long(q, k, v, o2, ws, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=b, seq=s, heads=h)
fns = [("product", base), ("long (csrc)", bounded)] + [(f"wide qb={qb} v={var}", mk(qb, var)) for qb, var in arms]
best = {n: 1e9 for n, _ in fns}
for r in range(4):
    for n, fn in fns:
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        best[n] = min(best[n], e0.elapsed_time(e1) * 1e3 / 10)
fl = 4.0 * b * h * s * s * 64
base()
for n, fn in fns[1:]:
    fn(); torch.cuda.synchronize()
    err = float((o.float() - o2.float()).norm() / o.float().norm())
    print(f"{n:22s} {best[n]:9.1f} us {fl / best[n] / 1e6:7.1f} TF/s  x{best['product'] / best[n]:.3f} vs product ({best['product']:.1f} us)  rel diff {err:.1e}", flush=True)

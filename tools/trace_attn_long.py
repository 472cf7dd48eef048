#!/usr/bin/env python3
"""Shader cycles per steady-state K/V tile of attn_long_kernel and the clock it ran at, from stamps inside the kernel
(s_memtime = shader clock, the 100 MHz wall clock beside it) around the steady loop of one wave of workgroup (3, 0).
Build the stamped library first:  make -C video-diffusion-pipeline-parallel_amd/csrc trace   (add EXTRA="-DLONG_TRACE=0
-DLONG_TRACE_ROWS" for marks inside the tile as well: each costs ~90 cycles and drains the LDS reads in flight).
usage: trace_attn_long.py [batch:seq:heads]      VDPP_HIP_LIB=<other stamped library>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd import hip
hip.LIB_PATH = os.path.abspath(os.environ.get("VDPP_HIP_LIB") or os.path.join(os.path.dirname(hip.LIB_PATH), "libsvdpipe_hip_trace.so"))
from vdpp_amd.hip import ops
b, s, h = map(int, (sys.argv[1] if len(sys.argv) > 1 else "14:9216:5").split(":"))
c = h * 64
qkv = torch.randn(b * s, 3 * c, device="cuda", dtype=torch.float16)
o = torch.empty(b * s, c, device="cuda", dtype=torch.float16)
need = ops.attn_long_ws_bytes(b, s, h)
ws = torch.zeros(need // 4 + 16, dtype=torch.int32, device="cuda")   # need: flag words + the flagged-waves word; + 16 stamps
for _ in range(3):
    ops.attn_spatial_long(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], o, ws, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=b, seq=s, heads=h)
torch.cuda.synchronize()
tr = ws[need // 4:].cpu().tolist()
n = s // 64 - 5
cyc, ticks = tr[14], tr[15]
print(f"steady loop of one wave: {cyc / n:.1f} shader cycles / tile, {ticks * 10.0 / n:.1f} ns / tile, clock {cyc / (ticks * 10.0):.3f} GHz")
if sum(tr[:14]):           # built with -DLONG_TRACE_ROWS as well: cycles up to each mark
    tot = sum(tr[:14])
    for k in range(14):
        print(f"mark {k:2d} {tr[k] / n:8.1f} cycles / tile  {100.0 * tr[k] / tot:5.1f} %")

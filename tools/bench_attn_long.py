#!/usr/bin/env python3
"""sp_attn_spatial_long_f16 (frozen-reference kernel + flagged second pass, csrc/attention_long.hip) against
sp_attn_spatial_f16, one process, product library.  usage: bench_attn_long.py batch:seq:heads[:mode] ...
mode: n = N(0,1) q/k/v (default); u = q, k x 2 (scores of +-40); l = every query has a late key far above its warm-up
maximum: every block overflows and is done again by the ordinary kernel (the worst case); h = half of the
(batch item, head) pairs as l."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd import hip
if os.environ.get("VDPP_HIP_LIB"):
    hip.LIB_PATH = os.path.abspath(os.environ["VDPP_HIP_LIB"])
from vdpp_amd.hip import ops

for spec in sys.argv[1:] or ["14:9216:5", "28:9216:5", "14:9216:5:l"]:
    parts = spec.split(":")
    b, s, h = map(int, parts[:3])
    mode = parts[3] if len(parts) > 3 else "n"
    c = h * 64
    torch.manual_seed(0)
    qkv = torch.randn(b * s, 3 * c, device="cuda", dtype=torch.float16)
    if mode == "u":
        qkv[:, :2 * c] *= 2.0
    if mode in "lh":
        x = qkv.view(b, s, 3 * h, 64)
        n = h if mode == "l" else max(1, h // 2)
        x[:, :, h:h + n] = (x[:, :, :n].roll(s // 3, dims=1) * 3.0)
    q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
    o = torch.empty(b * s, c, device="cuda", dtype=torch.float16)
    o2 = torch.empty_like(o)
    ws = torch.empty(ops.attn_long_ws_bytes(b, s, h), dtype=torch.uint8, device="cuda")
    kw = dict(ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=b, seq=s, heads=h)
    fns = [("ordinary", lambda: ops.attn_spatial(q, k, v, o, **kw)), ("long", lambda: ops.attn_spatial_long(q, k, v, o2, ws, **kw))]
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
    torch.cuda.synchronize()
    flagged = int(ws.view(torch.int32).sum())
    err = float((o.float() - o2.float()).norm() / o.float().norm())
    # fp32 reference on one (item, head)
    qq, kk, vv = [t[:s, :64].float() for t in (q, k, v)]
    ref = torch.softmax(qq @ kk.T * 0.125, dim=-1) @ vv
    e1_ = float((o[:s, :64].float() - ref).norm() / ref.norm()); e2_ = float((o2[:s, :64].float() - ref).norm() / ref.norm())
    print(f"{spec:16s} ordinary {best['ordinary']:8.1f} us {fl / best['ordinary'] / 1e6:7.1f} TF/s | long (incl. second-pass launch) {best['long']:8.1f} us "
          f"{fl / best['long'] / 1e6:7.1f} TF/s  x{best['ordinary'] / best['long']:.3f} | rel diff {err:.1e}; vs fp32: {e1_:.1e} / {e2_:.1e}; blocks flagged {flagged}/{b * h * s // 256}", flush=True)

#!/usr/bin/env python3
"""sp_gemm_f16 per kernel family on chosen shapes, interleaved rounds in one process (best of N per arm).
usage: bench_routes.py mode:m:n:cin[:g][:r][:r2] ...     g = GEGLU, r = one residual, r2 = two residuals
Arms: automatic choice, ping-pong (gemm_pp.hip), persistent-stream (gemm_ps.hip) with 256- and 192-row tiles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd.hip import ops

ARMS = [("auto", (0, 0, 0)), ("pp", (2, 0, 0)), ("ps256", (3, 256, 0)), ("ps192", (3, 192, 0)), ("ps128x320", (3, 128, 320)), ("ps256x192", (3, 256, 192))]
SPLITK_ARMS = [("small", (1, 0, 0)), ("128x64", (1, 128, 0)), ("split-K", (4, 0, 0)), ("auto+ws", (0, 0, 0))]   # m <= 6144


def run(spec, iters=20, rounds=4):
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
    # COLD=k: k sets of (A, output) buffers used in turn, so that a launch does not find the previous launch's operands in the
    # 256 MiB Infinity Cache (inside a forward a contraction's input was written by another kernel a while ago)
    nsets = int(os.environ.get("COLD", 1))
    a_sets = [torch.randn(m, cin, device=dev, dtype=torch.float16) for _ in range(nsets)]
    a = a_sets[0]
    wt = torch.randn(n, taps * cin, device=dev, dtype=torch.float16) * 0.02
    no = n // 2 if geglu else n
    out_sets = [torch.empty(m, no, device=dev, dtype=torch.float16) for _ in range(nsets)]
    out = out_sets[0]
    bias = torch.randn(n, device=dev)
    kw = dict(m=m, n=n, cin=cin, mode=mode, conv=conv, temporal=temporal, bias=bias, geglu=geglu)
    if "r" in flags or "r2" in flags:
        kw.update(res1=torch.randn(m, no, device=dev, dtype=torch.float16), r1scale=1.0)
    if "r2" in flags:
        kw.update(res2=torch.randn(m, no, device=dev, dtype=torch.float16), r2scale=0.5)
    arms = ARMS
    if m <= 6144:
        arms = SPLITK_ARMS
        need = ops.gemm_workspace_bytes(m=m, n=n, cin=cin, mode=mode)
        if need: kw.update(workspace=torch.empty(need, dtype=torch.uint8, device=dev))
    best = {name: 1e9 for name, _ in arms}
    ref = None
    for r in range(rounds):
        for name, route in arms:
            with ops.gemm_route(*route):
                for _ in range(2): ops.gemm(a, wt, out, **kw)
                torch.cuda.synchronize()
                if r == 0:
                    if ref is None: ref = out.float().clone()
                    else:
                        err = float((out.float() - ref).norm() / ref.norm())
                        assert err < 2e-3, (spec, name, err)
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for it in range(iters): ops.gemm(a_sets[it % nsets], wt, out_sets[it % nsets], **kw)
                e1.record(); torch.cuda.synchronize()
            best[name] = min(best[name], e0.elapsed_time(e1) * 1e3 / iters)
    fl = 2.0 * m * n * taps * cin
    print(f"{spec:30s} " + "  ".join(f"{name}: {best[name]:7.1f} us {fl / best[name] / 1e6:5.0f} TF" for name, _ in arms), flush=True)


if __name__ == "__main__":
    for s in sys.argv[1:]:
        run(s)

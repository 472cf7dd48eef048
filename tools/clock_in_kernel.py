#!/usr/bin/env python3
"""The shader clock the chip HOLDS inside the K loops of the heaviest contractions of a UNet forward, measured as
MI355X_MICROARCH.md prescribes ('DVFS give-back' item 6): s_memtime against the 100 MHz s_memrealtime around the loop, in
the experiments build (`make -C csrc exp`; the product build carries no stamp), after >= 2 s of back-to-back launches on
random data, median over workgroups.  Writes a JSON record that bench.py copies into `roofline.clock_ghz_in_kernel` (a
static record, like the PMC traffic: stamps cannot run inside the product kernels).
usage: clock_in_kernel.py OUT.json COMMIT"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import vdpp_amd  # noqa
from vdpp_amd import hip
hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), "libsvdpipe_hip_exp.so")
from vdpp_amd.hip import ops

# (name, mode, m, n, cin, residual): the three ping-pong shapes with the most time in a batched forward of two videos
SHAPES = [("conv3x3 level 0  258048 x 320 x 2880", 1, 258048, 320, 320, True),
          ("conv3x3 level 1   64512 x 640 x 5760", 1, 64512, 640, 640, True),
          ("FF2 level 0      258048 x 320 x 1280", 0, 258048, 320, 1280, True)]


def measure(name, mode, m, n, cin, resid):
    dev = "cuda"
    taps = 9 if mode == 1 else 1
    conv = None
    if mode == 1:
        h, w = (72, 128) if m == 258048 else (36, 64)
        conv = (m // (h * w), h, w, h, w, 1, 0)
    a = torch.randn(m, cin, device=dev, dtype=torch.float16)
    wt = torch.randn(n, taps * cin, device=dev, dtype=torch.float16) * 0.02
    out = torch.empty(m, n, device=dev, dtype=torch.float16)
    kw = dict(m=m, n=n, cin=cin, mode=mode, conv=conv, bias=torch.randn(n, device=dev))
    if resid:
        kw["res1"] = torch.randn(m, n, device=dev, dtype=torch.float16)
    hip.load().sp_gemm_set_route(2, 256, 0)          # the stamps live in gemm_pp.hip
    t0 = time.perf_counter()
    launches = 0
    while time.perf_counter() - t0 < 2.5:             # sustained load: the clock settles
        for _ in range(50):
            ops.gemm(a, wt, out, **kw)
        torch.cuda.synchronize()
        launches += 50
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm(a, wt, out, **kw); e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    nw = 16384
    buf = np.zeros((nw, 12), dtype=np.int64)
    assert hip.load().sp_debug_pp_trace(buf.ctypes.data_as(ctypes.c_void_p), nw) == 0
    hip.load().sp_gemm_set_route(0, 0, 0)
    used = buf[:, 0] > 0
    tmax = buf[used, 3].max()
    sel = used & (buf[:, 0] > tmax - int(us * 100 * 1.5))          # the workgroups of the last launch
    t = buf[sel].astype(np.float64)
    ghz = (t[:, 9] - t[:, 8]) / np.maximum(t[:, 2] - t[:, 1], 1) * 0.1   # shader cycles per 10 ns tick
    fl = 2.0 * m * n * taps * cin
    rec = {"shape": name, "us_per_launch": us, "tflops": fl / us / 1e6, "workgroups": int(sel.sum()),
           "clock_ghz_median": float(np.median(ghz)), "clock_ghz_p10": float(np.percentile(ghz, 10)),
           "clock_ghz_p90": float(np.percentile(ghz, 90)), "launches_before_the_stamped_one": launches}
    print(json.dumps(rec), flush=True)
    return rec


if __name__ == "__main__":
    out, commit = sys.argv[1], sys.argv[2]
    recs = [measure(*s) for s in SHAPES]
    w = [r["us_per_launch"] for r in recs]
    ghz = sum(r["clock_ghz_median"] * x for r, x in zip(recs, w)) / sum(w)
    json.dump({"commit": commit, "clock_ghz_in_kernel": ghz, "nominal_ghz": 2.4, "shapes": recs,
               "source": "tools/clock_in_kernel.py: s_memtime / s_memrealtime around the K loop of gemm_pp_kernel<256, 320> "
                         "(experiments build), after 2.5 s of back-to-back launches on random data, median over the "
                         "workgroups of one launch, time-weighted over the three heaviest ping-pong shapes"},
              open(out, "w"), indent=1)
    print(f"clock held inside the K loops: {ghz:.3f} GHz of 2.4 nominal -> {out}")

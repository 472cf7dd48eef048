#!/usr/bin/env python3
"""Which kernel family is fastest for each contraction shape INSIDE a forward (operands as cold as they are there), not in a
micro-benchmark that re-reads one cache-resident buffer: the benchmark's batched forward (two videos, 14 frames) is run launch by
launch with events around every sp_gemm_f16 call (ops.PROFILE), once per process-wide route, and the per-shape times are put
side by side.  A route only takes the shapes it supports; the rest keep the automatic choice (sp_gemm_set_route).
usage: route_tuner.py [--reps 3]"""
import argparse, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd import hip
from vdpp_amd.hip import ops
from vdpp_amd.models.svd_unet import StableVideoUNet

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--frames", type=int, default=14)
args = ap.parse_args()
ROUTES = [("auto", (0, 0, 0)), ("pp256", (2, 256, 0)), ("pp192", (2, 192, 0)), ("ps256", (3, 256, 0)), ("ps192", (3, 192, 0)),
          ("ps256x192", (3, 256, 192)), ("ps128x320", (3, 128, 320))]
dev = torch.device("cuda:0")
model = StableVideoUNet.from_random_init(StableVideoUNet._default_timestep_schedule(25), seed=0, device=dev)
torch.manual_seed(42)
model.set_dummy_conditioning(2, args.frames, 72, 128, dev)
lat = torch.randn(2, 4, args.frames, 72, 128, device=dev, dtype=torch.float16) * model.init_noise_sigma
best = collections.defaultdict(dict)       # shape -> route -> (us, kernel)
count = {}
with torch.no_grad():
    model(lat, 0); torch.cuda.synchronize()
    for rep in range(args.reps):
        for name, route in ROUTES:
            hip.load().sp_gemm_set_route(*route)
            ops.PROFILE = []
            model(lat, rep)
            torch.cuda.synchronize()
            prof, ops.PROFILE = ops.PROFILE, None
            acc = collections.defaultdict(lambda: [0.0, 0, None, 0.0])
            for kind, fl, e0, e1, nb, tag in prof:
                if kind != "gemm":
                    continue
                a = acc[tag[:5]]
                a[0] += e0.elapsed_time(e1) * 1e3; a[1] += 1; a[2] = tag[5]; a[3] += fl
            for shape, (us, n, kern, fl) in acc.items():
                count[shape] = (n, fl)
                old = best[shape].get(name)
                if old is None or us < old[0]:
                    best[shape][name] = (us, kern)
hip.load().sp_gemm_set_route(0, 0, 0)
rows = sorted(best.items(), key=lambda kv: -kv[1]["auto"][0])
print(f"{'shape [rows, cols, cin, mode, geglu]':40s} {'n':>3s} " + " ".join(f"{n:>10s}" for n, _ in ROUTES) + "   best (vs auto)")
tot = {n: 0.0 for n, _ in ROUTES}
for shape, r in rows:
    auto = r["auto"][0]
    line = f"{str(list(shape)):40s} {count[shape][0]:3d} " + " ".join(f"{r[n][0] / 1e3:10.3f}" for n, _ in ROUTES)
    bname = min(r, key=lambda k: r[k][0])
    gain = 1 - r[bname][0] / auto
    for n, _ in ROUTES:
        tot[n] += r[n][0]
    mark = f"   {bname} -{100 * gain:.1f} % ({r[bname][1]})" if gain > 0.03 and r[bname][1] != r["auto"][1] else ""
    print(line + mark)
print(f"{'sum of all contractions, ms':44s} " + " ".join(f"{tot[n] / 1e3:10.3f}" for n, _ in ROUTES))

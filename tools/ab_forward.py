#!/usr/bin/env python3
"""Same-box, same-process A/B of whole UNet forwards in the benchmark's configuration (micro-batches of two videos, two of
them in flight on two HIP streams): the arms alternate for ROUNDS rounds, best and median ms per video and forward.
usage: ab_forward.py [--lib exp] [--rounds 5] [--steps 6] ARM ARM ...
ARM = comma-separated settings: env:NAME=VALUE (process environment, e.g. env:SP_GEMM_DBG=256 with --lib exp) or
      unet:ATTR=0|1|INT (attribute of the SVDUNetHIP engine, e.g. unet:fold_groupnorm=0, unet:FF_CHUNK_BYTES=176160768) or
      route:R=BM (sp_gemm_set_route(R, BM, 0) for the whole forward, e.g. route:2=0 = ping-pong kernels only); "base" = nothing."""
import argparse, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--frames", type=int, default=14)
ap.add_argument("arms", nargs="+")
args = ap.parse_args()
import torch
import vdpp_amd  # noqa
from vdpp_amd import hip
if args.lib:
    hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), f"libsvdpipe_hip_{args.lib}.so")
from vdpp_amd.models.svd_unet import StableVideoUNet

dev = torch.device("cuda:0")
model = StableVideoUNet.from_random_init(StableVideoUNet._default_timestep_schedule(25), seed=0, device=dev)
B, S = 2, 2
torch.manual_seed(42)
model.set_dummy_conditioning(B, args.frames, 72, 128, dev)
lats = [torch.randn(B, 4, args.frames, 72, 128, device=dev, dtype=torch.float16) * model.init_noise_sigma for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
defaults = {}


def apply(arm):
    for k, v in defaults.items():
        setattr(model.unet, k, v)
    for name in [n for n in os.environ if n.startswith("SP_GEMM_")]:
        del os.environ[name]
    hip.load().sp_gemm_set_route(0, 0, 0)
    if arm == "base":
        return
    for item in arm.split(","):
        kind, rest = item.split(":", 1)
        name, val = rest.split("=", 1)
        if kind == "env":
            os.environ[name] = val
        elif kind == "route":
            hip.load().sp_gemm_set_route(int(name), int(val), 0)
        elif kind == "unet":
            defaults.setdefault(name, getattr(model.unet, name))
            setattr(model.unet, name, bool(int(val)) if val in ("0", "1") else int(val))
        else:
            raise SystemExit(f"unknown setting {item}")


def run():
    xs = list(lats)
    for s in range(args.steps):
        for i in range(S):
            with torch.cuda.stream(streams[i]):
                xs[i] = model(xs[i], s)
    for st in streams:
        st.synchronize()


times = {a: [] for a in args.arms}
with torch.no_grad():
    for a in args.arms:
        apply(a); run()
    torch.cuda.synchronize()
    for r in range(args.rounds):
        for a in args.arms:
            apply(a)
            torch.cuda.synchronize()
            t = time.perf_counter(); run(); torch.cuda.synchronize()
            times[a].append((time.perf_counter() - t) * 1e3 / (args.steps * B * S))
base = statistics.median(times[args.arms[0]])
for a in args.arms:
    med = statistics.median(times[a])
    print(f"{a:40s} best {min(times[a]):7.3f}  median {med:7.3f} ms per video and forward   ({100 * (med / base - 1):+.2f} % vs first arm)"
          f"   rounds: {' '.join(f'{t:.2f}' for t in times[a])}", flush=True)

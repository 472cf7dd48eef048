#!/usr/bin/env python3
"""Does it matter WHERE in its forward the second in-flight micro-batch is while the first one runs?  Two HIP streams run the
benchmark's batched forward (two videos each) back to back; the second stream starts `offset` of a forward late (a GPU-side
sleep in front of its first step).  Per-step times come from events on each stream; the steps in the middle of the run, where
both streams are busy, give the steady ms per video and forward for that phase offset.
usage: stream_offset.py [--steps 10] [offsets in ms ...]"""
import argparse, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd.models.svd_unet import StableVideoUNet

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("offsets", nargs="*", type=float, default=[0.0, 12.0, 24.0, 47.0, 70.0, 0.0])
args = ap.parse_args()
dev = torch.device("cuda:0")
model = StableVideoUNet.from_random_init(StableVideoUNet._default_timestep_schedule(25), seed=0, device=dev)
B = 2
torch.manual_seed(42)
model.set_dummy_conditioning(B, 14, 72, 128, dev)
lats = [torch.randn(B, 4, 14, 72, 128, device=dev, dtype=torch.float16) * model.init_noise_sigma for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
# calibrate torch.cuda._sleep (cycles of a fixed-rate counter) against wall time
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(100_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_ms = 100_000_000 / e0.elapsed_time(e1)


def run(offset_ms):
    xs = list(lats)
    marks = [[], []]
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            if i == 1 and offset_ms > 0:
                torch.cuda._sleep(int(offset_ms * cyc_per_ms))
            ev = torch.cuda.Event(enable_timing=True); ev.record(); marks[i].append(ev)
    for s in range(args.steps):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                xs[i] = model(xs[i], s)
                ev = torch.cuda.Event(enable_timing=True); ev.record(); marks[i].append(ev)
    torch.cuda.synchronize()
    per = [[marks[i][k].elapsed_time(marks[i][k + 1]) for k in range(args.steps)] for i in range(2)]
    mid = slice(2, args.steps - 2)             # both streams busy, away from start-up and tail
    return statistics.mean(per[0][mid] + per[1][mid]) / (2 * B) * 2, per   # ms per video and forward = step time / (2 streams' videos) * 2 streams... see below


with torch.no_grad():
    run(0.0)
    for off in args.offsets:
        # a step of one stream takes t while the other stream completes a step as well: 2 B videos per t
        m, per = run(off)
        t = statistics.mean(per[0][2:args.steps - 2] + per[1][2:args.steps - 2])
        print(f"offset {off:5.1f} ms: step time {t:7.2f} ms on each of two streams -> {t / (2 * B):6.2f} ms per video and forward   "
              f"(stream 0 steps: {' '.join(f'{x:.1f}' for x in per[0])})", flush=True)

"""Data-parallel comparison benchmark: every GPU runs ALL diffusion steps on its own share of the samples.

Counterpart of ``/root/reference/src/modes/benchmark_data_parallel.py`` (same flags, same ``BENCHMARK_JSON=`` keys): the
step pipeline is the product path (north star), this harness only supplies the number the reference reports beside it
(SURVEY.md section 8f item 4).  No communication happens during inference; timings are combined at the end with two
collectives (max wall clock, gathered per-rank first-sample times) instead of the reference's send/recv loop.
"""

from __future__ import annotations

import argparse
import json
import logging
import os
import time

import torch
import torch.distributed as dist

from ..distributed import finalize_distributed, init_distributed, resolve_backend

LOGGER = logging.getLogger(__name__)


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="Data parallel throughput benchmark")
    p.add_argument("--total-steps", type=int, default=28)
    p.add_argument("--num-samples", type=int, default=10)
    p.add_argument("--latent-channels", type=int, default=4)
    p.add_argument("--latent-frames", type=int, default=14)
    p.add_argument("--latent-height", type=int, default=40)
    p.add_argument("--latent-width", type=int, default=72)
    p.add_argument("--hidden-channels", type=int, default=64)
    p.add_argument("--warmup-samples", type=int, default=2)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--log-level", type=str, default="INFO")
    p.add_argument("--model", type=str, default="dummy", choices=["dummy", "svd"])
    p.add_argument("--model-id", type=str, default=None)
    p.add_argument("--backend", type=str, default="auto", choices=["auto", "gloo", "nccl"])
    p.add_argument("--init-method", type=str, default=None)
    p.add_argument("--guidance-scale", type=float, default=None)
    return p.parse_args(argv)


def main(argv=None) -> None:
    args = parse_args(argv)
    logging.basicConfig(level=getattr(logging, args.log_level.upper()),
                        format="%(asctime)s %(levelname)s %(name)s: %(message)s")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", 0)))
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    backend = resolve_backend(None if args.backend == "auto" else args.backend, simulator=False)
    init_distributed(backend=backend, rank=rank, world_size=world, init_method=args.init_method)

    use_svd = args.model == "svd"
    dtype = torch.float16 if use_svd else torch.float32
    if use_svd:
        from ..models.svd_unet import StableVideoUNet
        ts = StableVideoUNet._default_timestep_schedule(args.total_steps)
        model = (StableVideoUNet.from_pretrained(args.model_id, timesteps=ts, device=device) if args.model_id
                 else StableVideoUNet.from_random_init(ts, device=device))
        torch.manual_seed(args.seed)
        model.set_dummy_conditioning(1, args.latent_frames, args.latent_height, args.latent_width, device,
                                     guidance_scale=args.guidance_scale)
        scale = model.init_noise_sigma
    else:
        from ..models import DummyUNet
        torch.manual_seed(args.seed)
        model = DummyUNet(channels=args.latent_channels, hidden_channels=args.hidden_channels).to(device)
        scale = 1.0

    shape = torch.Size((1, args.latent_channels, args.latent_frames, args.latent_height, args.latent_width))
    total = args.warmup_samples + args.num_samples
    mine = [i for i in range(total) if i % world == rank]          # disjoint sample subsets
    measured_here = [i for i in mine if i >= args.warmup_samples]

    torch.cuda.synchronize(device)
    dist.barrier()
    t_start = time.perf_counter()
    ends = []
    with torch.no_grad():
        for i in mine:
            torch.manual_seed(args.seed + i)
            latent = torch.randn(shape, device=device, dtype=dtype) * scale
            for step in range(args.total_steps):
                latent = model(latent, step)
            torch.cuda.synchronize(device)
            ends.append(time.perf_counter())
    elapsed = torch.tensor([time.perf_counter() - t_start], dtype=torch.float64, device=device)
    first = torch.tensor([(ends[0] - t_start) if ends else 0.0], dtype=torch.float64, device=device)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    dist.all_reduce(first, op=dist.ReduceOp.MAX)
    counts = torch.tensor([len(measured_here)], dtype=torch.int64, device=device)
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)

    if rank == 0:
        per = [e - (t_start if k == 0 else ends[k - 1]) for k, e in enumerate(ends)]
        wall = float(elapsed.item())
        n_meas = int(counts.item())
        results = {
            "mode": "data_parallel", "world_size": world, "total_steps": args.total_steps,
            "steps_per_gpu": args.total_steps, "model": args.model, "num_samples_measured": n_meas,
            "warmup_samples": args.warmup_samples, "samples_per_rank": len(mine), "latent_shape": list(shape),
            "first_sample_time_s": round(float(first.item()), 4),
            "avg_sample_time_s": round(wall / max(len(mine), 1), 4),
            "throughput_samples_per_s": round(total / wall, 4) if wall > 0 else 0.0,
            "wall_clock_s": round(wall, 4),
            "per_sample_times_ms": [round(t * 1000, 2) for t in per],
        }
        print(f"BENCHMARK_JSON={json.dumps(results)}")
    finalize_distributed()


if __name__ == "__main__":
    main()

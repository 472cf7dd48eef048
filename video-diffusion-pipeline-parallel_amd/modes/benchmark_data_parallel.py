"""Data-parallel comparison benchmark: every GPU runs ALL diffusion steps on its own share of the samples.

Counterpart of ``/root/reference/src/modes/benchmark_data_parallel.py`` (same flags, same ``BENCHMARK_JSON=`` keys): the
step pipeline is the product path (north star), this harness only supplies the number the reference reports beside it
(SURVEY.md section 8f item 4).  No communication happens during inference; the per-rank (elapsed, count) pairs are
combined with one all_gather instead of the reference's send/recv loop.  ``measure`` holds the measurement itself.
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


def _sync(device: torch.device) -> None:
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def measure(model, *, shape, dtype, device, scale, rank: int, world: int, args) -> dict | None:
    """The reference's measurement (``benchmark_data_parallel.py:168-247``): EVERY rank runs ``warmup_samples`` untimed
    samples, then sync + barrier, then each rank times only its own contiguous share of the ``num_samples`` measured
    samples.  Throughput = measured samples of all ranks / the slowest rank's measured time; first / average sample
    time come from rank 0's measured samples.  Returns the ``BENCHMARK_JSON`` dict on rank 0, ``None`` elsewhere."""

    def one_sample(seed: int) -> None:
        torch.manual_seed(seed)
        latent = torch.randn(shape, device=device, dtype=dtype) * scale
        for step in range(args.total_steps):
            latent = model(latent, step)

    per_rank = -(-args.num_samples // world)                  # ceil: the last ranks may get fewer (or none)
    first, last = rank * per_rank, min((rank + 1) * per_rank, args.num_samples)

    _sync(device)
    dist.barrier()
    with torch.no_grad():
        for wi in range(args.warmup_samples):
            one_sample(args.seed + rank * 10000 + wi)
    _sync(device)
    dist.barrier()

    times: list[float] = []
    began = time.perf_counter()
    with torch.no_grad():
        for idx in range(first, last):
            t0 = time.perf_counter()
            one_sample(args.seed + idx)
            _sync(device)
            times.append(time.perf_counter() - t0)
    mine = torch.tensor([time.perf_counter() - began, float(max(last - first, 0))], dtype=torch.float64, device=device)
    dist.barrier()
    everyone = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(everyone, mine)      # (the reference funnels these two numbers to rank 0 with send/recv)
    else:
        everyone = [mine]
    if rank != 0:
        return None
    slowest = max(float(t[0]) for t in everyone)
    measured = sum(int(t[1]) for t in everyone)
    return {
        "mode": "data_parallel", "world_size": world, "total_steps": args.total_steps,
        "steps_per_gpu": args.total_steps, "model": args.model, "num_samples_measured": measured,
        "warmup_samples": args.warmup_samples, "samples_per_rank": per_rank, "latent_shape": list(shape),
        "first_sample_time_s": round(times[0], 4) if times else 0.0,
        "avg_sample_time_s": round(sum(times) / len(times), 4) if times else 0.0,
        "throughput_samples_per_s": round(measured / slowest, 4) if slowest > 0 else 0.0,
        "wall_clock_s": round(slowest, 4),
        "per_sample_times_ms": [round(t * 1000, 2) for t in times],
    }


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
    results = measure(model, shape=shape, dtype=dtype, device=device, scale=scale, rank=rank, world=world, args=args)
    if rank == 0:
        print(f"BENCHMARK_JSON={json.dumps(results)}")
    finalize_distributed()


if __name__ == "__main__":
    main()

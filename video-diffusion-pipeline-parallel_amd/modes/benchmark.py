"""Benchmark mode: pipeline-parallel throughput, printing the reference's ``BENCHMARK_JSON=`` line.

Counterpart of the pipeline branch of ``/root/reference/src/modes/benchmark.py`` (``:138-313``): warm-up +
measured samples, per-sample completion times on the last rank after a device sync, throughput =
n / sum(per-sample deltas) (``:254-267``), peak memory all_gather (``:240-249``).  The FSDP branch is out of
scope.  ``--model svd`` uses random weights of the exact architecture unless ``--model-id`` is a local dir.
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
from ..pipeline import LatentSpec, PipelineConfig, PipelineStage

LOGGER = logging.getLogger(__name__)


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="Pipeline parallel throughput benchmark")
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
    p.add_argument("--balanced", action="store_true")
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

    timesteps = list(range(args.total_steps - 1, -1, -1))
    shape = torch.Size((1, args.latent_channels, args.latent_frames, args.latent_height, args.latent_width))
    total = args.warmup_samples + args.num_samples
    torch.cuda.reset_peak_memory_stats(device)
    torch.cuda.synchronize(device)
    dist.barrier()
    ends: list[float] = []
    t_start = time.perf_counter()
    stage = PipelineStage(model, PipelineConfig(total_steps=args.total_steps, world_size=world, rank=rank,
                                                timesteps=timesteps, latent_spec=LatentSpec(shape, dtype, device),
                                                balanced=args.balanced))

    def supplier(i: int) -> torch.Tensor:
        torch.manual_seed(args.seed + i)
        return torch.randn(shape, device=device, dtype=dtype) * scale

    with torch.no_grad():
        for i in range(total):
            stage._more_samples_expected = i + 1 < total
            stage._process_single_latent(supplier(i) if rank == 0 else None, sample_idx=i)
            if rank == world - 1:
                torch.cuda.synchronize(device)
                ends.append(time.perf_counter())
    stage.drain()
    torch.cuda.synchronize(device)

    peak = torch.tensor([torch.cuda.max_memory_allocated(device)], dtype=torch.int64, device=device)
    gathered = [torch.zeros_like(peak) for _ in range(world)]
    if world > 1:
        dist.all_gather(gathered, peak)
    else:
        gathered = [peak]
    peaks = [float(t.item()) / 1e9 for t in gathered]

    if rank == world - 1:
        per = [e - (t_start if i == 0 else ends[i - 1]) for i, e in enumerate(ends)]
        measured = per[args.warmup_samples:]
        tot = sum(measured)
        results = {
            "world_size": world, "total_steps": args.total_steps,
            "steps_per_gpu": stage.step_range.count, "model": args.model, "fsdp": False,
            "num_samples_measured": args.num_samples, "warmup_samples": args.warmup_samples,
            "latent_shape": list(shape), "first_sample_time_s": round(per[0], 4) if per else 0.0,
            "avg_sample_time_s": round(tot / len(measured), 4) if measured else 0.0,
            "throughput_samples_per_s": round(len(measured) / tot, 4) if tot > 0 else 0.0,
            "per_sample_times_ms": [round(t * 1000, 2) for t in per],
            "peak_memory_gb_per_rank": [round(m, 3) for m in peaks], "max_peak_memory_gb": round(max(peaks), 3),
        }
        print(f"BENCHMARK_JSON={json.dumps(results)}")
    finalize_distributed()


if __name__ == "__main__":
    main()

"""Simulator mode: DummyUNet step pipeline on CPU/Gloo (or one GPU) under torchrun.

Counterpart of ``/root/reference/src/modes/simulator.py`` (same flags).  One deliberate difference:
the model is seeded (``--weight-seed``, default = ``--seed``) on EVERY rank before construction, so runs
are reproducible and all ranks hold identical weights; the reference builds the model unseeded
(``simulator.py:122``), which makes its CLI output differ from run to run.
"""

from __future__ import annotations

import argparse
import logging
import os

import torch

from ..distributed import finalize_distributed, init_distributed, resolve_backend
from ..models import DummyUNet
from ..pipeline import LatentSpec, run_single_latent

LOGGER = logging.getLogger(__name__)
_DTYPES = {"float32": torch.float32, "fp32": torch.float32, "float16": torch.float16, "fp16": torch.float16,
           "bfloat16": torch.bfloat16, "bf16": torch.bfloat16}


def str_to_dtype(name: str) -> torch.dtype:
    try:
        return _DTYPES[name.lower()]
    except KeyError:
        raise ValueError(f"Unsupported dtype '{name}'.") from None


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="Pipeline simulator mode")
    p.add_argument("--total-steps", type=int, default=28)
    p.add_argument("--rank", type=int, default=0)
    p.add_argument("--world-size", type=int, default=1)
    p.add_argument("--latent-batch", type=int, default=1)
    p.add_argument("--latent-channels", type=int, default=8)
    p.add_argument("--latent-frames", type=int, default=8)
    p.add_argument("--latent-height", type=int, default=32)
    p.add_argument("--latent-width", type=int, default=32)
    p.add_argument("--dtype", type=str, default="fp32")
    p.add_argument("--device", type=str, default="cpu")
    p.add_argument("--backend", type=str, default="auto", choices=["auto", "gloo", "nccl"])
    p.add_argument("--init-method", type=str, default=None)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--weight-seed", type=int, default=None)
    p.add_argument("--log-level", type=str, default="INFO")
    return p.parse_args(argv)


def main(argv=None) -> None:
    args = parse_args(argv)
    logging.basicConfig(level=getattr(logging, args.log_level.upper()),
                        format="%(asctime)s %(levelname)s %(name)s: %(message)s")
    rank = int(os.environ.get("RANK", args.rank))
    world = int(os.environ.get("WORLD_SIZE", args.world_size))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    backend = resolve_backend(None if args.backend == "auto" else args.backend, simulator=True)
    init_distributed(backend=backend, rank=rank, world_size=world, init_method=args.init_method)
    dtype = str_to_dtype(args.dtype)
    device = torch.device(f"cuda:{local_rank}") if args.device == "cuda" else torch.device(args.device)

    torch.manual_seed(args.seed if args.weight_seed is None else args.weight_seed)
    model = DummyUNet(channels=args.latent_channels).to(device)
    timesteps = list(reversed(range(args.total_steps)))
    shape = torch.Size((args.latent_batch, args.latent_channels, args.latent_frames, args.latent_height,
                        args.latent_width))
    spec = LatentSpec(shape=shape, dtype=dtype, device=device)
    latent = None
    if rank == 0:
        torch.manual_seed(args.seed)
        latent = torch.randn(shape, dtype=dtype).to(device)
    LOGGER.info("Simulator start rank=%s world_size=%s steps=%s backend=%s device=%s", rank, world,
                args.total_steps, backend, device)
    try:
        with torch.no_grad():
            final = run_single_latent(model=model, total_steps=args.total_steps, timesteps=timesteps,
                                      world_size=world, rank=rank, latent_spec=spec, input_latent=latent)
        if rank == world - 1 and final is not None:
            LOGGER.info("Final latent norm: %s", final.norm().item())
    finally:
        finalize_distributed()


if __name__ == "__main__":
    main()

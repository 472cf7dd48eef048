"""Production mode: SVD UNet step pipeline over RCCL under torchrun.

Counterpart of ``/root/reference/src/modes/production.py`` (same flags; ``--model-id`` must be a LOCAL
directory, ``--random-init`` runs the exact architecture with synthetic weights; ``--balanced`` allows
schedules that do not divide by the number of ranks, e.g. 25 steps on 8 GPUs).
"""

from __future__ import annotations

import argparse
import logging
import os

import torch

from ..distributed import finalize_distributed, init_distributed, resolve_backend
from ..pipeline import LatentSpec, run_pipeline_latents

LOGGER = logging.getLogger(__name__)


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="Pipeline production mode (SVD)")
    p.add_argument("--total-steps", type=int, default=28)
    p.add_argument("--num-samples", type=int, default=1)
    p.add_argument("--latent-frames", type=int, default=14)
    p.add_argument("--latent-height", type=int, default=72)
    p.add_argument("--latent-width", type=int, default=128)
    p.add_argument("--fps", type=int, default=6)
    p.add_argument("--motion-bucket-id", type=int, default=127)
    p.add_argument("--noise-aug-strength", type=float, default=0.02)
    p.add_argument("--guidance-scale", type=float, default=None)
    p.add_argument("--model-id", type=str, default="stabilityai/stable-video-diffusion-img2vid-xt")
    p.add_argument("--random-init", action="store_true")
    p.add_argument("--balanced", action="store_true")
    p.add_argument("--backend", type=str, default="auto", choices=["auto", "gloo", "nccl"])
    p.add_argument("--init-method", type=str, default=None)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--log-level", type=str, default="INFO")
    return p.parse_args(argv)


def main(argv=None) -> None:
    from ..models.svd_unet import StableVideoUNet

    args = parse_args(argv)
    logging.basicConfig(level=getattr(logging, args.log_level.upper()),
                        format="%(asctime)s %(levelname)s %(name)s: %(message)s")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    backend = resolve_backend(None if args.backend == "auto" else args.backend, simulator=False)
    init_distributed(backend=backend, rank=rank, world_size=world, init_method=args.init_method)

    timesteps = StableVideoUNet._default_timestep_schedule(args.total_steps)
    model = (StableVideoUNet.from_random_init(timesteps, device=device) if args.random_init
             else StableVideoUNet.from_pretrained(args.model_id, timesteps=timesteps, device=device))
    torch.manual_seed(args.seed)
    model.set_dummy_conditioning(1, args.latent_frames, args.latent_height, args.latent_width, device,
                                 fps=args.fps, motion_bucket_id=args.motion_bucket_id,
                                 noise_aug_strength=args.noise_aug_strength, guidance_scale=args.guidance_scale)
    shape = torch.Size((1, 4, args.latent_frames, args.latent_height, args.latent_width))
    spec = LatentSpec(shape=shape, dtype=torch.float16, device=device)

    def supplier(i: int) -> torch.Tensor:
        torch.manual_seed(args.seed + i)
        return torch.randn(shape, device=device, dtype=torch.float16) * model.init_noise_sigma

    try:
        with torch.no_grad():
            outs = run_pipeline_latents(model, total_steps=args.total_steps,
                                        timesteps=list(range(args.total_steps - 1, -1, -1)), world_size=world,
                                        rank=rank, latent_spec=spec, num_samples=args.num_samples,
                                        input_supplier=supplier if rank == 0 else None, balanced=args.balanced)
        if outs:
            torch.cuda.synchronize(device)
            for i, o in enumerate(outs):
                LOGGER.info("sample %d final latent norm: %s", i, o.float().norm().item())
    finally:
        finalize_distributed()


if __name__ == "__main__":
    main()

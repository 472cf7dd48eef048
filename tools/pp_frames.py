#!/usr/bin/env python3
"""The step pipeline WITH its last edge stage: WORLD_SIZE ranks (sharing the cards that exist, Gloo hand-off) push K toy
videos through a small SVD UNet and a small temporal VAE decoder; models/edge_stages.py::FrameEmitter decodes every finished
latent on the rank where it finished (ring: rank (i mod N) - 1; chain: the last rank), beside the UNet steps, on a stream of
its own -- no finished latent is ever forwarded.  Every rank
writes {sample index: frames} of what IT decoded to --out-dir/rank<r>.pt.  Run at world sizes 1, 2, 3 the union of the
files must be bit-identical, and equal to decode_latents of the plain loop's latents (tests/test_modes_gpu.py).
usage: pp_frames.py --out-dir DIR [--samples 5] [--schedule rotate|chain|ring] [--concurrent 2] [--no-spread]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd.distributed import finalize_distributed, init_distributed
from vdpp_amd.models.edge_stages import FrameEmitter
from vdpp_amd.models.svd_unet import StableVideoUNet
from vdpp_amd.models.unet_hip import SVDUNetHIP
from vdpp_amd.models.unet_spec import UNetConfig, random_state_dict
from vdpp_amd.models.vae_hip import TemporalDecoderHIP, VAEDecoderConfig
from vdpp_amd.models.vae_hip import random_state_dict as vae_random_state_dict
from vdpp_amd.pipeline import LatentSpec, PipelineConfig, PipelineStage


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out-dir", required=True)
    ap.add_argument("--samples", type=int, default=5)
    ap.add_argument("--schedule", default="rotate", choices=["chain", "rotate", "ring"])
    ap.add_argument("--concurrent", type=int, default=2)
    ap.add_argument("--no-spread", action="store_true")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--micro-batch", type=int, default=1)
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dev = torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', rank)) % max(1, torch.cuda.device_count())}")
    torch.cuda.set_device(dev)
    if world > 1:
        init_distributed(backend="gloo", rank=rank, world_size=world, init_method=None)
    frames, h, w, mb = 3, 8, 16, args.micro_batch
    ucfg = UNetConfig.tiny(64)
    unet = SVDUNetHIP(ucfg, random_state_dict(ucfg, seed=0, dtype=torch.float16), dev)
    model = StableVideoUNet(unet=unet, timesteps=StableVideoUNet._default_timestep_schedule(args.steps))
    torch.manual_seed(42)
    model.set_dummy_conditioning(mb, frames, h, w, dev)
    vcfg = VAEDecoderConfig.tiny(64)
    dec = TemporalDecoderHIP(vcfg, vae_random_state_dict(vcfg, seed=19), dev)
    spec = LatentSpec(shape=torch.Size((mb, 4, frames, h, w)), dtype=torch.float16, device=dev)
    cfg = PipelineConfig(total_steps=args.steps, timesteps=list(range(args.steps)), world_size=world, rank=rank,
                         latent_spec=spec, balanced=True, rotate=args.schedule == "rotate" and world > 1,
                         ring=args.schedule == "ring", concurrent_samples=args.concurrent,
                         async_comm=True if world > 1 and args.schedule != "ring" else None)
    stage = PipelineStage(model, cfg)
    emitter = FrameEmitter(dec, stage, frames, spread=not args.no_spread, keep="all")

    def supplier(i):
        g = torch.Generator().manual_seed(1000 + i)
        return (torch.randn(spec.shape, generator=g) * model.init_noise_sigma).half().to(dev)

    with torch.no_grad():
        out = stage.run_many(args.samples, input_supplier=supplier if (rank == 0 or args.schedule == "ring") else None)
        stage.drain()
        mine = emitter.finish(args.samples)
    torch.cuda.synchronize()
    os.makedirs(args.out_dir, exist_ok=True)
    torch.save({"frames": {i: t.cpu() for i, t in mine.items()}, "stats": emitter.stats,
                "latents": [t.cpu() for t in out] if out is not None else None}, os.path.join(args.out_dir, f"rank{rank}.pt"))
    print(f"[rank {rank}] decoded samples {sorted(mine)} stats {emitter.stats}", flush=True)
    if world > 1:
        finalize_distributed()


if __name__ == "__main__":
    main()

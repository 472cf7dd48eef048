#!/usr/bin/env python3
"""BASELINE configs 3/4 on their own workload, minus the transport hardware: the real SVD UNet (1.52 B parameters,
random init, seeded) on the benchmark latent, 25 steps, pushed through the step pipeline by WORLD_SIZE ranks that
share the cards that exist (Gloo hand-off between the processes; RCCL refuses two ranks on one device).  The last rank
writes the finished latents to --out; run at world sizes 1, 2, 4 the files must be bit-identical (every kernel is
deterministic and a stage boundary only moves the fp16 latent).  Launch with torch.distributed.run.
usage: pp_equivalence.py --out FILE [--samples 3] [--schedule chain|rotate|ring] [--concurrent 2] [--frames 14]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd.distributed import finalize_distributed, init_distributed
from vdpp_amd.models.svd_unet import StableVideoUNet
from vdpp_amd.pipeline import LatentSpec, PipelineConfig, PipelineStage


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--samples", type=int, default=3)
    ap.add_argument("--schedule", default="rotate", choices=["chain", "rotate", "ring"])
    ap.add_argument("--concurrent", type=int, default=2)
    ap.add_argument("--frames", type=int, default=14)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--async-link", action="store_true",
                    help="hand latents over through pipeline._SideStreamLink (pre-posted irecv, isend behind an event, fresh "
                         "receive buffers) instead of blocking send/recv; over Gloo the link orders on the host")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dev = torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', rank)) % max(1, torch.cuda.device_count())}")
    torch.cuda.set_device(dev)
    init_distributed(backend="gloo", rank=rank, world_size=world, init_method=None)
    ts = StableVideoUNet._default_timestep_schedule(args.steps)
    model = StableVideoUNet.from_random_init(ts, seed=0, device=dev)
    torch.manual_seed(42)
    model.set_dummy_conditioning(1, args.frames, 72, 128, dev)
    spec = LatentSpec(shape=torch.Size((1, 4, args.frames, 72, 128)), dtype=torch.float16, device=dev)
    cfg = PipelineConfig(total_steps=args.steps, timesteps=list(range(args.steps)), world_size=world, rank=rank,
                         latent_spec=spec, balanced=True, rotate=args.schedule == "rotate", ring=args.schedule == "ring",
                         concurrent_samples=args.concurrent, async_comm=True if (args.async_link and world > 1) else None)
    stage = PipelineStage(model, cfg)
    if os.environ.get("PPEQ_DEBUG"):        # checksum of what enters every step (first video only: concurrent 1)
        inner = model.forward
        seen = {"n": 0}

        def logged(latent, step):
            if seen["n"] < args.steps:
                print(f"[rank {rank}] step {step:2d} in  {float(latent.float().abs().sum()):.6e}  "
                      f"cond {float(model._image_latents.float().abs().sum()):.6e} "
                      f"emb {float(model._image_embeddings.float().abs().sum()):.6e}", flush=True)
            seen["n"] += 1
            return inner(latent, step)
        model.forward = logged

    def supplier(i):
        g = torch.Generator().manual_seed(1000 + i)
        return (torch.randn(spec.shape, generator=g) * model.init_noise_sigma).half().to(dev)

    with torch.no_grad():
        out = stage.run_many(args.samples, input_supplier=supplier if (rank == 0 or args.schedule == "ring") else None)
    torch.cuda.synchronize()
    if rank == world - 1:
        assert out is not None and len(out) == args.samples
        torch.save([t.cpu() for t in out], args.out)
        print(f"world {world} schedule {args.schedule}: {len(out)} latents -> {args.out}", flush=True)
    stage.drain()
    print(f"[rank {rank}] transport {stage.transport}", flush=True)
    finalize_distributed()


if __name__ == "__main__":
    main()

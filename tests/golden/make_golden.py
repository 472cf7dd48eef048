"""Mint golden vectors by running the REFERENCE's own code (``/root/reference/src``) on seeded inputs.

Run once in the build container (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Outputs (committed, data only – inputs, weights and expected outputs):
  dummy_c8h16.npz   DummyUNet(8,16), latent (1,8,8,32,32) fp32, 8 steps [7..0]  (BASELINE cfg 1)
                    final latent for world_size 1 and for world_size 2 over Gloo (bit-identical),
                    plus per-step norms and the stage-boundary latent (every step's latent for the small case).
  dummy_c4h64.npz   DummyUNet(4,64), latent (2,4,3,8,12), 4 steps [3..0]
  svd_step.npz      StableVideoUNet.forward (ref svd_unet.py:351-439) with a deterministic stub
                    UNet, (1,4,6,8,8), fp32 + fp16, CFG off / 3.0, steps 0, 12, 24 of 25.

The reference CLI seeds nothing before building the model (simulator.py:122 vs :90), so the
harness seeds on every rank before construction.

``StableVideoUNet.__init__`` imports ``diffusers.EulerDiscreteScheduler`` (absent offline); only
its sigma/timestep table is needed, so a stand-in module exposing the restated table
(``oracle/euler_sched.py``) is injected.  The table is therefore an INPUT of the fixture; the
arithmetic under test (scale, concat, permute, CFG, Euler update) is the reference's.
"""

from __future__ import annotations

import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

WEIGHT_SEED = 1234
INPUT_SEED = 42


def _dummy_worker(rank, ws, channels, hidden, shape, steps, init_file, out_file):
    from src.distributed.setup import finalize_distributed, init_distributed
    from src.models.dummy_unet import DummyUNet
    from src.pipeline.pipeline import LatentSpec, run_single_latent

    torch.set_num_threads(2)
    if ws > 1:
        init_distributed(backend="gloo", rank=rank, world_size=ws, init_method=f"file://{init_file}")
    torch.manual_seed(WEIGHT_SEED)
    model = DummyUNet(channels=channels, hidden_channels=hidden)
    torch.manual_seed(INPUT_SEED)
    x = torch.randn(shape)
    spec = LatentSpec(shape=torch.Size(shape), dtype=torch.float32, device=torch.device("cpu"))
    with torch.no_grad():
        out = run_single_latent(model, total_steps=steps, timesteps=list(reversed(range(steps))),
                                world_size=ws, rank=rank, latent_spec=spec,
                                input_latent=x if rank == 0 else None)
    if rank == ws - 1:
        torch.save(out, out_file)
    if ws > 1:
        finalize_distributed()


def mint_dummy(name, channels, hidden, shape, steps, world_sizes):
    import torch.multiprocessing as mp
    from src.models.dummy_unet import DummyUNet

    torch.manual_seed(WEIGHT_SEED)
    model = DummyUNet(channels=channels, hidden_channels=hidden)
    torch.manual_seed(INPUT_SEED)
    x = torch.randn(shape)
    ts = list(reversed(range(steps)))
    per_step = []
    lat = x
    with torch.no_grad():
        for s in ts:
            lat = model(lat, s)
            per_step.append(lat.numpy().copy())
    finals = {}
    for ws in world_sizes:
        with tempfile.TemporaryDirectory() as td:
            out_file = os.path.join(td, "out.pt")
            args = (ws, channels, hidden, shape, steps, os.path.join(td, "init"), out_file)
            if ws == 1:
                _dummy_worker(0, *args)
            else:
                mp.spawn(_dummy_worker, args=args, nprocs=ws, join=True)
            finals[ws] = torch.load(out_file).numpy()
    for ws, f in finals.items():
        assert f.tobytes() == per_step[-1].tobytes(), f"ws={ws} differs from the plain loop"
    arrays = {f"param.{k}": v.numpy() for k, v in model.state_dict().items()}
    arrays.update(input=x.numpy(), timesteps=np.asarray(ts, np.int32),
                  per_step_norm=np.asarray([np.linalg.norm(p.astype(np.float64)) for p in per_step]),
                  midpoint=per_step[steps // 2 - 1],  # what rank 0 sends to rank 1 at world_size 2
                  final=per_step[-1], world_sizes=np.asarray(world_sizes, np.int32))
    if sum(p.nbytes for p in per_step) <= 256 * 1024:
        arrays["per_step"] = np.stack(per_step)
    np.savez_compressed(os.path.join(HERE, name), **arrays)
    print(name, "final norm", float(np.linalg.norm(per_step[-1])), "ws", world_sizes, "bit-identical")


def _stub_unet(sample, timestep, encoder_hidden_states, added_time_ids, return_dict=False):
    """Deterministic stand-in for the UNet: (B,F,8,H,W) -> (B,F,4,H,W), touches every input."""
    a, b = sample[:, :, :4].float(), sample[:, :, 4:].float()
    t = float(timestep)
    bias = encoder_hidden_states.float().mean() * 0.25 + added_time_ids.float().sum() * 1e-3
    out = 0.7 * a - 0.2 * b + 0.05 * t + 0.1 * torch.tanh(a * b) + bias
    return (out.to(sample.dtype),)


def mint_svd_step():
    from oracle import euler_sched

    class EulerDiscreteScheduler:  # stand-in: table only (see module docstring)
        def __init__(self, **kw):
            self.kw = kw

        def set_timesteps(self, n):
            sig = euler_sched.karras_sigmas(n, self.kw["sigma_min"], self.kw["sigma_max"])
            self.sigmas = torch.from_numpy(sig)
            self.timesteps = torch.from_numpy(euler_sched.continuous_timesteps(sig))

    sys.modules["diffusers"] = types.SimpleNamespace(EulerDiscreteScheduler=EulerDiscreteScheduler)
    from src.models.svd_unet import StableVideoUNet

    n = 25
    shape = (1, 4, 6, 8, 8)
    g = torch.Generator().manual_seed(7)
    emb = torch.randn(1, 1, 1024, generator=g)
    img = torch.randn(shape, generator=g)
    arrays = {"image_embeddings": emb.numpy(), "image_latents": img.numpy()}
    for dt_name, dt in (("fp32", torch.float32), ("fp16", torch.float16)):
        stub = torch.nn.Module()
        stub.forward = _stub_unet
        model = StableVideoUNet(unet=stub, timesteps=StableVideoUNet._default_timestep_schedule(n),
                                dtype=dt)
        arrays["sigmas"] = model.sigmas.numpy()
        arrays["scheduler_timesteps"] = model.scheduler_timesteps.numpy()
        arrays["init_noise_sigma"] = np.float64(model.init_noise_sigma)
        for gs_name, gs in (("nocfg", None), ("cfg3", 3.0)):
            model.set_conditioning(emb, img, guidance_scale=gs, num_frames=shape[2])
            arrays[f"added_time_ids.{dt_name}"] = model._added_time_ids.float().numpy()
            for step in (0, 12, 24):
                x = (torch.randn(shape, generator=g) * float(model.sigmas[step] + 1)).to(dt)
                y = model(x, step)
                arrays[f"in.{dt_name}.{gs_name}.{step}"] = x.float().numpy()
                arrays[f"out.{dt_name}.{gs_name}.{step}"] = y.float().numpy()
    np.savez_compressed(os.path.join(HERE, "svd_step.npz"), **arrays)
    print("svd_step.npz", len(arrays), "arrays; sigma0", arrays["sigmas"][0], "sigma1",
          arrays["sigmas"][1], "t0", arrays["scheduler_timesteps"][0])


if __name__ == "__main__":
    mint_dummy("dummy_c8h16.npz", 8, 16, (1, 8, 8, 32, 32), 8, [1, 2])
    mint_dummy("dummy_c4h64.npz", 4, 64, (2, 4, 3, 8, 12), 4, [1, 2, 4])
    mint_svd_step()

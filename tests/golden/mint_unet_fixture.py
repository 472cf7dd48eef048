#!/usr/bin/env python3
"""DORMANT pin for the two parity-unpinned oracles (oracle/svd_unet_ref.py, oracle/euler_sched.py).

`diffusers` (declared >=0.20.0, run at 0.36.0 by the reference authors, /root/reference/EXPERIMENT_REPORT.md:36-41) is not
installed in the build container and is not vendored in the reference, so the SVD UNet arithmetic and the Karras sigma
table are restatements of absent third-party code.  This script does NOT fetch or vendor anything.  Run it only in a
container where `import diffusers` already works; it then mints

    tests/golden/unet_tiny_diffusers.npz   a tiny-config UNetSpatioTemporalConditionModel (same topology as UNetConfig.tiny(64):
                                           block_out_channels (64,128,256,256), heads (1,2,4,4), cross dim 128): its
                                           seeded state_dict, seeded inputs and the fp32 output of ONE forward
    tests/golden/euler_tables_diffusers.npz the EulerDiscreteScheduler sigma / timestep tables for N = 25 and 30 built
                                           exactly as /root/reference/src/models/svd_unet.py:77-102 builds them

and tests/test_oracle_cpu.py::test_unet_oracle_matches_diffusers_fixture / test_schedule_matches_diffusers_fixture
(skipped while the files are absent) then pin the oracles against them.  Every assumption those oracles make that the
reference itself cannot confirm is listed in DESIGN.md section 4.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main() -> int:
    try:
        from diffusers import EulerDiscreteScheduler, UNetSpatioTemporalConditionModel
    except Exception as exc:  # noqa: BLE001
        print(f"diffusers is not importable here ({exc!r}); nothing minted (the oracles stay parity-unpinned).")
        return 1
    import diffusers

    # ---- scheduler tables (ref svd_unet.py:77-102)
    tabs = {}
    for n in (25, 30):
        sch = EulerDiscreteScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                     num_train_timesteps=1000, prediction_type="v_prediction", timestep_spacing="leading",
                                     timestep_type="continuous", steps_offset=1, use_karras_sigmas=True,
                                     sigma_min=0.002, sigma_max=700.0, interpolation_type="linear", rescale_betas_zero_snr=False)
        sch.set_timesteps(n)
        tabs[f"sigmas.{n}"] = sch.sigmas.float().numpy()
        tabs[f"timesteps.{n}"] = sch.timesteps.float().numpy()
        tabs[f"init_noise_sigma.{n}"] = np.float32(float(sch.init_noise_sigma))
    np.savez_compressed(os.path.join(HERE, "euler_tables_diffusers.npz"), diffusers_version=diffusers.__version__, **tabs)

    # ---- tiny UNet (topology of UNetConfig.tiny(64))
    torch.manual_seed(20261004)
    unet = UNetSpatioTemporalConditionModel(
        sample_size=None, in_channels=8, out_channels=4,
        down_block_types=("CrossAttnDownBlockSpatioTemporal",) * 3 + ("DownBlockSpatioTemporal",),
        up_block_types=("UpBlockSpatioTemporal",) + ("CrossAttnUpBlockSpatioTemporal",) * 3,
        block_out_channels=(64, 128, 256, 256), addition_time_embed_dim=32, projection_class_embeddings_input_dim=96,
        layers_per_block=2, cross_attention_dim=128, transformer_layers_per_block=1, num_attention_heads=(1, 2, 4, 4),
        num_frames=4).eval().float()
    with torch.no_grad():                      # the zero-initialised / constant parameters would hide mistakes
        for name, p in unet.named_parameters():
            if p.dim() <= 1:
                p.copy_(torch.randn_like(p) * 0.2 + (1.0 if name.endswith("norm.weight") or ".norm" in name and name.endswith("weight") else 0.0))
    g = torch.Generator().manual_seed(7)
    frames, h, w = 4, 16, 24
    sample = torch.randn(1, frames, 8, h, w, generator=g)
    ctx = torch.randn(1, 1, 128, generator=g)
    ids = torch.tensor([[5.0, 127.0, 0.02]])
    t = torch.tensor(1.25)
    with torch.no_grad():
        out = unet(sample, t, encoder_hidden_states=ctx, added_time_ids=ids, return_dict=False)[0]
    np.savez_compressed(os.path.join(HERE, "unet_tiny_diffusers.npz"), diffusers_version=diffusers.__version__,
                        sample=sample.numpy(), timestep=t.numpy(), encoder_hidden_states=ctx.numpy(),
                        added_time_ids=ids.numpy(), out=out.float().numpy(),
                        **{"param." + k: v.float().numpy() for k, v in unet.state_dict().items()})
    print("minted unet_tiny_diffusers.npz and euler_tables_diffusers.npz with diffusers", diffusers.__version__)
    return 0


if __name__ == "__main__":
    sys.exit(main())

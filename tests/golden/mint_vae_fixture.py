#!/usr/bin/env python3
"""DORMANT pin for the parity-unpinned VAE oracle (oracle/vae_temporal_decoder_ref.py: TemporalDecoderRef + EncoderRef).

`diffusers` (declared >=0.20.0, run at 0.36.0 by the reference authors, /root/reference/EXPERIMENT_REPORT.md:36-41) is not
installed in the build container and is not vendored in the reference; the reference reaches it at
/root/reference/scripts/generate_video_demo.py:139-148 (`vae.encode(image).latent_dist.mode()`) and :181
(`vae.decode(chunk, num_frames=...).sample`).  This script does NOT fetch or vendor anything.  Run it only in a container
where `import diffusers` already works; it then mints

    tests/golden/vae_tiny_diffusers.npz   a tiny-config AutoencoderKLTemporalDecoder (block_out_channels (32,64,128,128),
                                          the topology of VAEDecoderConfig.tiny(32)): its seeded state_dict, a seeded latent
                                          chunk (B*F, 4, H, W) with num_frames, the decoder's fp32 output, a seeded image
                                          and the encoder's `latent_dist.mode()`

and tests/test_oracle_cpu.py::test_vae_oracle_matches_diffusers_fixture (skipped while the file is absent) pins both halves
of the oracle against it.  The assumptions the oracle makes that the reference itself cannot confirm are listed in its
header and in DESIGN.md section 4.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main() -> int:
    try:
        from diffusers import AutoencoderKLTemporalDecoder
    except Exception as exc:  # noqa: BLE001
        print(f"diffusers is not importable here ({exc!r}); nothing minted (the VAE oracle stays parity-unpinned).")
        return 1
    import diffusers

    torch.manual_seed(20261005)
    vae = AutoencoderKLTemporalDecoder(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4,
                                       block_out_channels=(32, 64, 128, 128), layers_per_block=2, latent_channels=4,
                                       sample_size=64, scaling_factor=0.18215, force_upcast=True).eval().float()
    with torch.no_grad():                      # constant norm parameters / zero biases would hide mistakes
        for name, p in vae.named_parameters():
            if p.dim() <= 1:
                p.copy_(torch.randn_like(p) * 0.2 + (1.0 if "norm" in name and name.endswith("weight") else 0.0))
    g = torch.Generator().manual_seed(11)
    frames, h, w = 3, 8, 12
    z = torch.randn(2 * frames, 4, h, w, generator=g)                     # two videos of three frames
    image = torch.randn(2, 3, 8 * h, 8 * w, generator=g).clamp(-1, 1)
    with torch.no_grad():
        dec = vae.decode(z, num_frames=frames).sample
        mode = vae.encode(image).latent_dist.mode()
    np.savez_compressed(os.path.join(HERE, "vae_tiny_diffusers.npz"), diffusers_version=diffusers.__version__,
                        z=z.numpy(), num_frames=np.int64(frames), decoded=dec.float().numpy(), image=image.numpy(),
                        latent_mode=mode.float().numpy(),
                        **{"param." + k: v.float().numpy() for k, v in vae.state_dict().items()})
    print("minted vae_tiny_diffusers.npz with diffusers", diffusers.__version__)
    return 0


if __name__ == "__main__":
    sys.exit(main())

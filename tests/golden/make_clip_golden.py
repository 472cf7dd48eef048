"""Mint the CLIP image-encoder golden vector by running the reference's own dependency.

The reference computes its image embeddings with ``transformers.CLIPVisionModelWithProjection``
(``/root/reference/scripts/generate_video_demo.py:108-112, 248-254``); transformers is installed in this image, so the
vector is minted by the real third-party code, not by a restatement:

    python tests/golden/make_clip_golden.py        # transformers version is recorded in the file

Output (committed, data only): ``clip_tiny.npz`` -- config, the state_dict (fp16-representable fp32 values) of a small
CLIPVisionModelWithProjection (hidden 128, 3 layers, 2 heads of 64, 56x56 image, 14x14 patches, projection 64,
quick_gelu), ``pixel_values`` (2,3,56,56) and the resulting ``image_embeds`` / ``last_hidden_state`` in fp32.
``tests/test_oracle_cpu.py`` re-runs transformers against it (version drift shows there), ``tests/test_clip_gpu.py``
checks the HIP engine against the stored outputs.
"""

from __future__ import annotations

import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = dict(hidden_size=128, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2, image_size=56, patch_size=14,
           projection_dim=64, hidden_act="quick_gelu", layer_norm_eps=1e-5)


def build(cfg_kwargs=CFG, seed=20260404):
    import transformers
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

    torch.manual_seed(seed)
    model = CLIPVisionModelWithProjection(CLIPVisionConfig(**cfg_kwargs)).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("bias") or "norm" in name:       # transformers initialises these to 0 / 1: make them matter
                p.add_(0.1 * torch.randn(p.shape, generator=g))
        for p in model.parameters():
            p.copy_(p.half().float())
    return model, transformers.__version__


def main():
    model, version = build()
    g = torch.Generator().manual_seed(7)
    px = torch.randn(2, 3, CFG["image_size"], CFG["image_size"], generator=g).half().float()
    with torch.no_grad():
        out = model(px)
    arrays = {"w:" + k: v.numpy() for k, v in model.state_dict().items()}
    arrays.update(pixel_values=px.numpy(), image_embeds=out.image_embeds.numpy(),
                  last_hidden_state=out.last_hidden_state.numpy(), transformers_version=np.array(version),
                  config=np.array(repr(sorted(CFG.items()))))
    path = os.path.join(HERE, "clip_tiny.npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB), transformers {version}")


if __name__ == "__main__":
    main()

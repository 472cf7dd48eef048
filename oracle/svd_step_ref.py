"""ORACLE (test infrastructure): one ``StableVideoUNet.forward`` step, numpy/torch fp32.

Follows ``/root/reference/src/models/svd_unet.py:351-439`` line by line:
scale input (``:382``), optional sequential CFG with per-frame guidance (``:385-411``),
single pass otherwise (``:413-422``), permute back (``:425``), fp32 v-prediction Euler update
(``:428-437``), cast to storage dtype (``:439``).  ``unet`` is any callable with the diffusers
keyword signature; pinned against the reference's own forward via
``tests/golden/svd_step_*.npz`` (minted by ``tests/golden/make_golden.py``).
"""

from __future__ import annotations

import torch


def svd_step(unet, latent, step, *, sigmas, timesteps, image_embeddings, image_latents,
             added_time_ids, guidance_scale=None, dtype=torch.float16):
    n = len(timesteps)
    if not (0 <= step < n):
        raise ValueError(f"Step {step} out of range [0, {n})")
    sigma = sigmas[step]
    sigma_next = sigmas[step + 1]
    t = timesteps[step]
    scaled = latent / ((sigma ** 2 + 1) ** 0.5)

    def run(img_lat, emb):
        x = torch.cat([scaled, img_lat], dim=1).permute(0, 2, 1, 3, 4).to(dtype)
        return unet(sample=x, timestep=t, encoder_hidden_states=emb,
                    added_time_ids=added_time_ids, return_dict=False)[0]

    if guidance_scale is not None and guidance_scale > 1.0:
        nf = latent.shape[2]
        gs = torch.linspace(1.0, guidance_scale, nf).view(1, nf, 1, 1, 1).to(latent.device, dtype)
        un = run(torch.zeros_like(image_latents), torch.zeros_like(image_embeddings))
        co = run(image_latents, image_embeddings)
        eps = un + gs * (co - un)
    else:
        eps = run(image_latents, image_embeddings)
    eps = eps.permute(0, 2, 1, 3, 4)

    x = latent.float()
    e = eps.float()
    s = sigma.float() if torch.is_tensor(sigma) else torch.tensor(float(sigma))
    x0 = e * (-s / (s ** 2 + 1) ** 0.5) + x / (s ** 2 + 1)
    d = (x - x0) / s
    dt = float(sigma_next) - float(sigma)
    return (x + d * dt).to(dtype)

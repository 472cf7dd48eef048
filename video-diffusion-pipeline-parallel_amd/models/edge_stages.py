"""What the first and the last pipeline stage do around the denoising steps, as the reference's demo arranges it
(``/root/reference/scripts/generate_video_demo.py``): ``encode_image`` on rank 0 (``:92-151``) and ``decode_latents`` on
the last rank (``:154-195``), on the HIP engines of ``clip_hip.py`` / ``vae_hip.py``.

Same argument meaning and return values as the script's functions, minus what is host-side image handling there: the
``CLIPImageProcessor`` output (``pixel_values``) and the normalised image tensor are arguments instead of a PIL image.
"""

from __future__ import annotations

import torch

from .clip_hip import CLIPVisionHIP
from .vae_hip import ImageEncoderHIP, TemporalDecoderHIP


def encode_image(pixel_values: torch.Tensor, image_tensor: torch.Tensor, image_encoder: CLIPVisionHIP,
                 vae_encoder: ImageEncoderHIP, num_frames: int, noise: torch.Tensor | None = None,
                 noise_aug_strength: float = 0.0) -> tuple[torch.Tensor, torch.Tensor]:
    """``(image_embeddings (B,1,1024), image_latents (B,4,F,H/8,W/8))`` as ``StableVideoUNet.set_conditioning`` takes
    them.  ``pixel_values``: ``feature_extractor(images=image).pixel_values`` (ref ``:108-109``); ``image_tensor``:
    ``Normalize([0.5],[0.5])(ToTensor(image))`` (ref ``:117-124``).  Noise augmentation in PIXEL space (ref ``:126-129``):
    the reference draws ``randn_like`` on its device; here the caller passes the ``noise`` tensor it wants added (device
    RNG streams differ between platforms), scaled by ``noise_aug_strength``.  The latents are the distribution's mode,
    NOT multiplied by the scaling factor (ref ``:139-142``)."""
    dev = image_encoder.device
    emb = image_encoder(pixel_values.to(dev, torch.float16).contiguous()).unsqueeze(1)
    img = image_tensor.to(dev, torch.float16)
    if noise is not None and noise_aug_strength > 0:
        img = torch.add(img, noise.to(dev, torch.float16), alpha=noise_aug_strength)    # input preparation, once per video
    latents = vae_encoder.encode_image_latents(img.contiguous(), num_frames)
    return emb, latents


def decode_latents(latents: torch.Tensor, vae: TemporalDecoderHIP, num_frames: int,
                   decode_chunk_size: int = 14) -> torch.Tensor:
    """(B, 4, F, H, W) latents -> (B, 3, F, 8H, 8W) fp32 frames (ref ``:154-195``)."""
    return vae.decode_latents(latents.to(vae.device, torch.float16).contiguous(), num_frames,
                              decode_chunk_size=decode_chunk_size)

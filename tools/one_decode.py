#!/usr/bin/env python3
"""Temporal VAE decode of one 14 x 576 x 1024 video, NREP times (default 3: one warm + two), for rocprofv3 --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd.models.vae_hip import TemporalDecoderHIP, VAEDecoderConfig, random_state_dict
dev = torch.device("cuda:0")
cfg = VAEDecoderConfig.svd()
dec = TemporalDecoderHIP(cfg, random_state_dict(cfg, seed=0), dev)
lat = (torch.randn((1, 4, 14, 72, 128), device=dev) * cfg.scaling_factor).half()
with torch.no_grad():
    for _ in range(int(os.environ.get("NREP", 3))):
        vid = dec.decode_latents(lat, 14, decode_chunk_size=14)
torch.cuda.synchronize()
print("done", tuple(vid.shape))

"""ORACLE (test infrastructure, never shipped on the product path).

Plain-PyTorch fp32 restatement of the temporal VAE decoder the reference's demo script calls on the last
stage (``/root/reference/scripts/generate_video_demo.py:154-195``: ``decode_latents`` ->
``vae.decode(chunk, num_frames=chunk.shape[0]).sample`` with ``vae`` =
``diffusers.AutoencoderKLTemporalDecoder`` loaded at ``:251-262``).  SURVEY.md 8f-3.

diffusers (declared ``>=0.20.0`` in ``/root/reference/requirements.txt:2``, run at 0.36.0 per
``/root/reference/EXPERIMENT_REPORT.md:39``) is NOT vendored in the reference and not installed here, and no
checkpoint exists in this image, so this file restates the published architecture
(``models/autoencoders/autoencoder_kl_temporal_decoder.py::TemporalDecoder``, ``unets/unet_3d_blocks.py::
MidBlockTemporalDecoder / UpBlockTemporalDecoder``, ``resnet.py::SpatioTemporalResBlock / AlphaBlender``,
``attention_processor.py::Attention``) with the SVD VAE config: ``block_out_channels (128, 256, 512, 512)``,
``layers_per_block 2``, ``latent_channels 4``, ``out_channels 3``, ``scaling_factor 0.18215``, ``force_upcast true``.

**Parity unpinned**: the reference holds no golden vector at the VAE boundary (``/root/reference/tests`` never
touches it; the demo script only writes an mp4).  Anchored on the call site above and on the diffusers parameter
naming (``decoder.*``), so that a real ``vae`` state_dict loads into ``TemporalDecoderRef`` with ``strict=True``.

What this restatement ASSUMES about diffusers 0.36 (each is a place where a pinned run could disagree):
 (1) ``conv_in`` 3x3 -> mid block (resnet, single-head attention, resnet) -> four up blocks of three
     SpatioTemporalResBlocks (the first three followed by nearest x2 + 3x3 conv) -> GroupNorm(32, eps 1e-6) + SiLU ->
     ``conv_out`` 3x3 -> ``time_conv_out`` Conv3d (3,1,1) over the frames of the chunk; no ``post_quant_conv``.
 (2) SpatioTemporalResBlock here has NO time embedding (``temb_channels=None``), spatial GroupNorm eps 1e-6,
     temporal GroupNorm eps 1e-5; spatial block adds a 1x1 ``conv_shortcut`` when the widths differ.
 (3) AlphaBlender: ``merge_strategy="learned"``, ``switch_spatial_to_temporal_mix=True``:
     ``alpha = 1 - sigmoid(mix_factor)``; out = ``alpha * spatial + (1 - alpha) * temporal``.
 (4) mid-block attention: ``Attention(query_dim=512, heads=1, dim_head=512, bias=True, norm_num_groups=32,
     eps=1e-6, residual_connection=True)`` over the H*W tokens of EACH frame: GroupNorm on (B*F, C, HW),
     q/k/v projections with bias, softmax(q.k^T / sqrt(512)), out projection with bias, + residual; no dropout.
 (5) the temporal GroupNorms / (3,1,1) convolutions see exactly the frames of one ``decode`` call (the reference
     decodes ``decode_chunk_size=14`` frames per call), zero padding in time.
 (6) ``decode_latents`` divides by ``scaling_factor`` first and returns ``(B, 3, F, 8H, 8W)`` in fp32.

The ENCODER half (``vae.encode(image).latent_dist.mode()``, ref ``generate_video_demo.py:117-145``) is restated at the
bottom of this file (``EncoderRef`` / ``encode_image_latents``) under the same status; its assumptions:
 (7) ``Encoder``: ``conv_in`` 3x3 -> four DownEncoderBlock2D of two ResnetBlock2D (eps 1e-6, no time embedding, 1x1
     shortcut when the widths differ), the first three followed by Downsample2D(padding=0): ``F.pad(x, (0,1,0,1))`` then
     a stride-2 3x3 convolution WITHOUT padding -> UNetMidBlock2D (resnet, the same single-head attention as (4),
     resnet) -> GroupNorm(32, eps 1e-6) + SiLU -> ``conv_out`` to ``2*latent_channels``; then ``quant_conv`` 1x1.
 (8) ``latent_dist.mode()`` is the mean = the first ``latent_channels`` of those; NO ``scaling_factor`` on this path.

Only ``tests/`` and ``bench.py``'s optional decode check import this.
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class VAEDecoderConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Sequence[int] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_groups: int = 32
    scaling_factor: float = 0.18215

    @staticmethod
    def svd() -> "VAEDecoderConfig":
        return VAEDecoderConfig()

    @staticmethod
    def tiny(c: int = 64) -> "VAEDecoderConfig":
        """Same topology, narrow channels (the mid attention keeps ONE head of width 4c)."""
        return VAEDecoderConfig(block_out_channels=(c, 2 * c, 4 * c, 4 * c))


class _Res2D(nn.Module):
    def __init__(self, cin, cout, eps, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class _ResT(nn.Module):
    def __init__(self, c, eps, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, c, eps=eps)
        self.conv1 = nn.Conv3d(c, c, (3, 1, 1), padding=(1, 0, 0))
        self.norm2 = nn.GroupNorm(groups, c, eps=eps)
        self.conv2 = nn.Conv3d(c, c, (3, 1, 1), padding=(1, 0, 0))

    def forward(self, x):                       # (B, C, F, H, W)
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return x + h


class _Mixer(nn.Module):
    def __init__(self):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.tensor([0.0]))


class SpatioTemporalResBlock(nn.Module):
    """diffusers ``SpatioTemporalResBlock(temb_channels=None, eps=1e-6, temporal_eps=1e-5, merge_strategy="learned",
    switch_spatial_to_temporal_mix=True)``."""

    def __init__(self, cin, cout, groups):
        super().__init__()
        self.spatial_res_block = _Res2D(cin, cout, 1e-6, groups)
        self.temporal_res_block = _ResT(cout, 1e-5, groups)
        self.time_mixer = _Mixer()

    def forward(self, x, frames):               # (B*F, C, H, W)
        s = self.spatial_res_block(x)
        bf, c, h, w = s.shape
        s5 = s.reshape(bf // frames, frames, c, h, w).permute(0, 2, 1, 3, 4)
        t5 = self.temporal_res_block(s5)
        alpha = 1.0 - torch.sigmoid(self.time_mixer.mix_factor).to(s.dtype)     # switch_spatial_to_temporal_mix
        out = alpha * s5 + (1.0 - alpha) * t5
        return out.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


class _Attention(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q = nn.Linear(c, c)
        self.to_k = nn.Linear(c, c)
        self.to_v = nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c)])

    def forward(self, x):                       # (B*F, C, H, W), ONE head of width C
        n, c, h, w = x.shape
        t = self.group_norm(x.reshape(n, c, h * w)).transpose(1, 2)       # (n, HW, C)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        p = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(c), dim=-1)
        o = self.to_out[0](p @ v)
        return x + o.transpose(1, 2).reshape(n, c, h, w)


class _Mid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(c, c, groups), SpatioTemporalResBlock(c, c, groups)])
        self.attentions = nn.ModuleList([_Attention(c, groups)])

    def forward(self, x, frames):
        x = self.resnets[0](x, frames)
        x = self.attentions[0](x)
        return self.resnets[1](x, frames)


class _Upsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Up(nn.Module):
    def __init__(self, cin, cout, layers, upsample, groups):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(cin if i == 0 else cout, cout, groups)
                                      for i in range(layers)])
        self.upsamplers = nn.ModuleList([_Upsample(cout)]) if upsample else None

    def forward(self, x, frames):
        for r in self.resnets:
            x = r(x, frames)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class TemporalDecoderRef(nn.Module):
    """``AutoencoderKLTemporalDecoder.decoder`` (parameter names as in diffusers: load a real vae state_dict's
    ``decoder.*`` entries with the prefix stripped)."""

    def __init__(self, cfg: VAEDecoderConfig):
        super().__init__()
        self.cfg = cfg
        ch = list(cfg.block_out_channels)
        g = cfg.norm_groups
        self.conv_in = nn.Conv2d(cfg.latent_channels, ch[-1], 3, padding=1)
        self.mid_block = _Mid(ch[-1], g)
        rev = list(reversed(ch))
        ups, prev = [], rev[0]
        for i, c in enumerate(rev):
            ups.append(_Up(prev, c, cfg.layers_per_block + 1, i != len(rev) - 1, g))
            prev = c
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(g, ch[0], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, padding=1)
        self.time_conv_out = nn.Conv3d(cfg.out_channels, cfg.out_channels, (3, 1, 1), padding=(1, 0, 0))

    def forward(self, z, num_frames):           # z (B*F, 4, H, W) -> (B*F, 3, 8H, 8W)
        x = self.conv_in(z)
        x = self.mid_block(x, num_frames)
        for up in self.up_blocks:
            x = up(x, num_frames)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        bf, c, h, w = x.shape
        x5 = x.reshape(bf // num_frames, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        x5 = self.time_conv_out(x5)
        return x5.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


def decode_latents(latents, decoder: TemporalDecoderRef, num_frames: int, decode_chunk_size: int = 14):
    """``/root/reference/scripts/generate_video_demo.py:154-195`` with ``vae.decode`` = ``decoder``:
    (B, 4, F, H, W) -> (B, 3, F, 8H, 8W) fp32; frames are decoded ``decode_chunk_size`` at a time."""
    lat = latents.permute(0, 2, 1, 3, 4)
    b = lat.shape[0]
    lat = lat.flatten(0, 1) / decoder.cfg.scaling_factor
    frames = []
    with torch.no_grad():
        for i in range(0, lat.shape[0], decode_chunk_size):
            chunk = lat[i:i + decode_chunk_size]
            frames.append(decoder(chunk, chunk.shape[0]))
    out = torch.cat(frames, dim=0)
    out = out.reshape(b, num_frames, *out.shape[1:]).permute(0, 2, 1, 3, 4)
    return out.float()


def decoder_flops(cfg: VAEDecoderConfig, frames: int, h: int, w: int) -> float:
    """Multiply-add FLOPs (2 per MAC) of one ``decoder(z, frames)`` call at latent size h x w."""
    ch = list(cfg.block_out_channels)
    total = 0.0
    px = frames * h * w

    def res(cin, cout, px):
        f = 2.0 * px * 9 * cin * cout + 2.0 * px * 9 * cout * cout + 2 * 2.0 * px * 3 * cout * cout
        if cin != cout:
            f += 2.0 * px * cin * cout
        return f

    c = ch[-1]
    total += 2.0 * px * 9 * cfg.latent_channels * c
    total += 2 * res(c, c, px)
    total += 4 * 2.0 * px * c * c + 2 * 2.0 * frames * (h * w) ** 2 * c
    rev = list(reversed(ch))
    prev = rev[0]
    for i, co in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            total += res(prev if j == 0 else co, co, px)
        if i != len(rev) - 1:
            px *= 4
            total += 2.0 * px * 9 * co * co
        prev = co
    total += 2.0 * px * 9 * ch[0] * cfg.out_channels + 2.0 * px * 3 * cfg.out_channels ** 2
    return total


# ------------------------------------------------------------------------------------------------ encoder half
class _Downsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class _Down(nn.Module):
    def __init__(self, cin, cout, layers, downsample, groups):
        super().__init__()
        self.resnets = nn.ModuleList([_Res2D(cin if i == 0 else cout, cout, 1e-6, groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([_Downsample(cout)]) if downsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class _Mid2D(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([_Res2D(c, c, 1e-6, groups), _Res2D(c, c, 1e-6, groups)])
        self.attentions = nn.ModuleList([_Attention(c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class EncoderRef(nn.Module):
    """``AutoencoderKLTemporalDecoder.encoder`` + ``.quant_conv`` (load the checkpoint's ``encoder.*`` entries under
    ``encoder.`` and ``quant_conv.*`` as they are)."""

    def __init__(self, cfg: VAEDecoderConfig, in_channels: int = 3):
        super().__init__()
        self.cfg = cfg
        ch = list(cfg.block_out_channels)
        g = cfg.norm_groups
        enc = nn.Module()
        enc.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
        downs, prev = [], ch[0]
        for i, c in enumerate(ch):
            downs.append(_Down(prev, c, cfg.layers_per_block, i != len(ch) - 1, g))
            prev = c
        enc.down_blocks = nn.ModuleList(downs)
        enc.mid_block = _Mid2D(ch[-1], g)
        enc.conv_norm_out = nn.GroupNorm(g, ch[-1], eps=1e-6)
        enc.conv_out = nn.Conv2d(ch[-1], 2 * cfg.latent_channels, 3, padding=1)
        self.encoder = enc
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)

    def forward(self, x):                       # (B, 3, H, W) -> moments (B, 2*latent, H/8, W/8)
        e = self.encoder
        x = e.conv_in(x)
        for d in e.down_blocks:
            x = d(x)
        x = e.mid_block(x)
        x = e.conv_out(F.silu(e.conv_norm_out(x)))
        return self.quant_conv(x)


def encode_image_latents(image, encoder: EncoderRef, num_frames: int):
    """``/root/reference/scripts/generate_video_demo.py:139-148``: ``vae.encode(image).latent_dist.mode()`` (the mean,
    no scaling factor) repeated over the frames: (B, 3, H, W) -> (B, latent, F, H/8, W/8)."""
    with torch.no_grad():
        mean = encoder(image)[:, :encoder.cfg.latent_channels]
    return mean.unsqueeze(2).repeat(1, 1, num_frames, 1, 1)

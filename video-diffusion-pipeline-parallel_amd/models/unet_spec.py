"""Architecture description of the SVD UNet (``UNetSpatioTemporalConditionModel``) and its parameter
inventory in diffusers' state_dict naming.

The reference loads this network from diffusers (``/root/reference/src/models/svd_unet.py:129-136``);
this module only *describes* it (channel plan, module tree, parameter names/shapes) so that

  * a real ``unet`` state_dict (``diffusion_pytorch_model.safetensors``) can be mapped onto the HIP
    engine, and
  * the benchmark can create random weights of exactly that architecture (no network for checkpoints).
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Iterator, Sequence

import torch


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 8
    out_channels: int = 4
    block_out_channels: Sequence[int] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    num_attention_heads: Sequence[int] = (5, 10, 20, 20)
    cross_attention_dim: int = 1024
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 768
    norm_groups: int = 32
    down_has_attn: Sequence[bool] = (True, True, True, False)

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @staticmethod
    def svd() -> "UNetConfig":
        """SVD img2vid and SVD-XT share this UNet configuration."""
        return UNetConfig()

    @staticmethod
    def tiny(c: int = 64) -> "UNetConfig":
        """Same topology with narrow channels (head dim stays 64) for fast parity tests."""
        return UNetConfig(
            block_out_channels=(c, 2 * c, 4 * c, 4 * c),
            num_attention_heads=(c // 64, 2 * c // 64, 4 * c // 64, 4 * c // 64),
            cross_attention_dim=128,
            addition_time_embed_dim=32,
            projection_class_embeddings_input_dim=96,
        )


# ----------------------------------------------------------------------------------------------
# module tree -> (name, shape, fan_in) ; fan_in = 0 marks "ones", -1 marks "zeros", -2 mix_factor
# ----------------------------------------------------------------------------------------------
def _linear(p, cin, cout, bias=True):
    yield f"{p}.weight", (cout, cin), cin
    if bias:
        yield f"{p}.bias", (cout,), cin


def _conv(p, cin, cout, k):
    yield f"{p}.weight", (cout, cin) + tuple(k), cin * math.prod(k)
    yield f"{p}.bias", (cout,), cin * math.prod(k)


def _norm(p, c):
    yield f"{p}.weight", (c,), 0
    yield f"{p}.bias", (c,), -1


def _resblock(p, cin, cout, temb):
    s, t = f"{p}.spatial_res_block", f"{p}.temporal_res_block"
    yield from _norm(f"{s}.norm1", cin)
    yield from _conv(f"{s}.conv1", cin, cout, (3, 3))
    yield from _linear(f"{s}.time_emb_proj", temb, cout)
    yield from _norm(f"{s}.norm2", cout)
    yield from _conv(f"{s}.conv2", cout, cout, (3, 3))
    if cin != cout:
        yield from _conv(f"{s}.conv_shortcut", cin, cout, (1, 1))
    yield from _norm(f"{t}.norm1", cout)
    yield from _conv(f"{t}.conv1", cout, cout, (3, 1, 1))
    yield from _linear(f"{t}.time_emb_proj", temb, cout)
    yield from _norm(f"{t}.norm2", cout)
    yield from _conv(f"{t}.conv2", cout, cout, (3, 1, 1))
    yield f"{p}.time_mixer.mix_factor", (1,), -2


def _attention(p, dim, ctx_dim):
    yield from _linear(f"{p}.to_q", dim, dim, bias=False)
    yield from _linear(f"{p}.to_k", ctx_dim, dim, bias=False)
    yield from _linear(f"{p}.to_v", ctx_dim, dim, bias=False)
    yield from _linear(f"{p}.to_out.0", dim, dim)


def _ff(p, dim, dim_out=None):
    yield from _linear(f"{p}.net.0.proj", dim, dim * 8)
    yield from _linear(f"{p}.net.2", dim * 4, dim_out or dim)


def _transformer(p, c, cross):
    yield from _norm(f"{p}.norm", c)
    yield from _linear(f"{p}.proj_in", c, c)
    b = f"{p}.transformer_blocks.0"
    yield from _norm(f"{b}.norm1", c)
    yield from _attention(f"{b}.attn1", c, c)
    yield from _norm(f"{b}.norm2", c)
    yield from _attention(f"{b}.attn2", c, cross)
    yield from _norm(f"{b}.norm3", c)
    yield from _ff(f"{b}.ff", c)
    t = f"{p}.temporal_transformer_blocks.0"
    yield from _norm(f"{t}.norm_in", c)
    yield from _ff(f"{t}.ff_in", c)
    yield from _norm(f"{t}.norm1", c)
    yield from _attention(f"{t}.attn1", c, c)
    yield from _norm(f"{t}.norm2", c)
    yield from _attention(f"{t}.attn2", c, cross)
    yield from _norm(f"{t}.norm3", c)
    yield from _ff(f"{t}.ff", c)
    yield from _linear(f"{p}.time_pos_embed.linear_1", c, 4 * c)
    yield from _linear(f"{p}.time_pos_embed.linear_2", 4 * c, c)
    yield f"{p}.time_mixer.mix_factor", (1,), -2
    yield from _linear(f"{p}.proj_out", c, c)


def up_block_plan(cfg: UNetConfig):
    """[(in_ch, out_ch, prev_ch, has_attn, heads, upsample)] for up_blocks.0..3 (diffusers get_up_block)."""
    rev = list(cfg.block_out_channels)[::-1]
    heads = list(cfg.num_attention_heads)[::-1]
    attn = list(cfg.down_has_attn)[::-1]
    plan, out_ch = [], rev[0]
    for i in range(len(rev)):
        prev, out_ch = out_ch, rev[i]
        plan.append((rev[min(i + 1, len(rev) - 1)], out_ch, prev, attn[i], heads[i], i != len(rev) - 1))
    return plan


def param_inventory(cfg: UNetConfig) -> Iterator[tuple]:
    boc = list(cfg.block_out_channels)
    temb, cross = cfg.time_embed_dim, cfg.cross_attention_dim
    yield from _conv("conv_in", cfg.in_channels, boc[0], (3, 3))
    yield from _linear("time_embedding.linear_1", boc[0], temb)
    yield from _linear("time_embedding.linear_2", temb, temb)
    yield from _linear("add_embedding.linear_1", cfg.projection_class_embeddings_input_dim, temb)
    yield from _linear("add_embedding.linear_2", temb, temb)
    ch = boc[0]
    for i, cout in enumerate(boc):
        for j in range(cfg.layers_per_block):
            yield from _resblock(f"down_blocks.{i}.resnets.{j}", ch if j == 0 else cout, cout, temb)
            if cfg.down_has_attn[i]:
                yield from _transformer(f"down_blocks.{i}.attentions.{j}", cout, cross)
        if i != len(boc) - 1:
            yield from _conv(f"down_blocks.{i}.downsamplers.0.conv", cout, cout, (3, 3))
        ch = cout
    yield from _resblock("mid_block.resnets.0", ch, ch, temb)
    yield from _transformer("mid_block.attentions.0", ch, cross)
    yield from _resblock("mid_block.resnets.1", ch, ch, temb)
    layers = cfg.layers_per_block + 1
    for i, (in_ch, out_ch, prev, attn, _heads, ups) in enumerate(up_block_plan(cfg)):
        for j in range(layers):
            skip = in_ch if j == layers - 1 else out_ch
            rin = prev if j == 0 else out_ch
            yield from _resblock(f"up_blocks.{i}.resnets.{j}", rin + skip, out_ch, temb)
            if attn:
                yield from _transformer(f"up_blocks.{i}.attentions.{j}", out_ch, cross)
        if ups:
            yield from _conv(f"up_blocks.{i}.upsamplers.0.conv", out_ch, out_ch, (3, 3))
    yield from _norm("conv_norm_out", boc[0])
    yield from _conv("conv_out", boc[0], cfg.out_channels, (3, 3))


def random_state_dict(cfg: UNetConfig, seed: int = 0, device="cpu", dtype=torch.float16) -> dict:
    """Random weights of the exact architecture: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like
    ``nn.Linear`` / ``nn.Conv*`` defaults, norm scale 1 (+small jitter), norm bias small."""
    gen = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape, fan in param_inventory(cfg):
        if fan > 0:
            bound = 1.0 / math.sqrt(fan)
            t = (torch.rand(shape, generator=gen, device=device, dtype=torch.float32) * 2 - 1) * bound
        elif fan == 0:
            t = 1.0 + 0.1 * (torch.rand(shape, generator=gen, device=device) - 0.5)
        elif fan == -1:
            t = 0.1 * (torch.rand(shape, generator=gen, device=device) - 0.5)
        else:  # AlphaBlender mix_factor
            t = torch.rand(shape, generator=gen, device=device) - 0.5
        sd[name] = t.to(dtype)
    return sd


def param_count(cfg: UNetConfig) -> int:
    return sum(math.prod(s) for _, s, _ in param_inventory(cfg))


# ----------------------------------------------------------------------------------------------
# FLOP model of one forward (bench.py prices the MFMA roofline with it; SURVEY.md 8(d) inventory)
# ----------------------------------------------------------------------------------------------
def contractions(cfg: UNetConfig, frames: int, h: int, w: int) -> Iterator[tuple]:
    """Every contraction of one UNet forward as ``(kind, m, n, k)`` = an ``[m][k] x [n][k]^T`` product (2*m*n*k
    FLOPs), plus ``("attn_s" | "attn_t", batch_heads, seq, 64)`` for the attention products (4*bh*seq^2*64).
    ``kind`` in conv3x3 / tconv / conv1x1 / linear / cross_qo; ``cross_qo`` are the Q and out projections of the
    single-token cross-attention modules, which cannot change the result and which the HIP engine does not run."""
    temb = cfg.time_embed_dim

    def resblock(cin, cout, hh, ww):
        m = frames * hh * ww
        yield "conv3x3", m, cout, 9 * cin
        yield "conv3x3", m, cout, 9 * cout
        yield "tconv", m, cout, 3 * cout
        yield "tconv", m, cout, 3 * cout
        if cin != cout:
            yield "conv1x1", m, cout, cin
        yield "linear", 1, cout, temb              # time_emb_proj of the spatial and of the temporal block
        yield "linear", 1, cout, temb

    def transformer(c, hh, ww):
        m = frames * hh * ww
        yield "linear", m, c, c                    # proj_in
        yield "linear", m, c, c                    # proj_out
        for _ in range(2):                         # spatial block, temporal block
            yield "linear", m, 3 * c, c            # q, k, v
            yield "linear", m, c, c                # self-attention out
            yield "cross_qo", m, c, c              # cross-attention q
            yield "cross_qo", m, c, c              # cross-attention out
            yield "linear", m, 8 * c, c            # GEGLU in
            yield "linear", m, c, 4 * c            # ff out
        yield "linear", m, 8 * c, c                # temporal ff_in
        yield "linear", m, c, 4 * c
        yield "attn_s", frames * (c // 64), hh * ww, 64
        yield "attn_t", hh * ww * (c // 64), frames, 64

    boc = list(cfg.block_out_channels)
    hh, ww = h, w
    yield "conv3x3", frames * h * w, boc[0], 9 * cfg.in_channels
    ch = boc[0]
    for i, cout in enumerate(boc):
        for j in range(cfg.layers_per_block):
            yield from resblock(ch if j == 0 else cout, cout, hh, ww)
            if cfg.down_has_attn[i]:
                yield from transformer(cout, hh, ww)
        ch = cout
        if i != len(boc) - 1:
            hh, ww = (hh + 1) // 2, (ww + 1) // 2
            yield "conv3x3", frames * hh * ww, cout, 9 * cout          # stride-2 downsampler
    yield from resblock(ch, ch, hh, ww)
    yield from transformer(ch, hh, ww)
    yield from resblock(ch, ch, hh, ww)
    layers = cfg.layers_per_block + 1
    for in_ch, out_ch, prev, attn, _heads, ups in up_block_plan(cfg):
        for j in range(layers):
            skip = in_ch if j == layers - 1 else out_ch
            yield from resblock((prev if j == 0 else out_ch) + skip, out_ch, hh, ww)
            if attn:
                yield from transformer(out_ch, hh, ww)
        if ups:
            hh, ww = hh * 2, ww * 2
            yield "conv3x3", frames * hh * ww, out_ch, 9 * out_ch       # conv after the nearest x2 upsample
    yield "conv3x3", frames * h * w, cfg.out_channels, 9 * boc[0]


def forward_flops(cfg: UNetConfig, frames: int, h: int, w: int, count_cross_attn_qo: bool = True) -> dict:
    """Algorithmic FLOPs of one forward by op class and in ``total`` (44.69 TFLOP for SVD at 14 x 72 x 128;
    43.08 with ``count_cross_attn_qo=False`` = what the HIP engine executes)."""
    tot = {"conv3x3": 0.0, "tconv": 0.0, "conv1x1": 0.0, "linear": 0.0, "attn_s": 0.0, "attn_t": 0.0, "cross_qo": 0.0}
    for kind, m, n, k in contractions(cfg, frames, h, w):
        tot[kind] += 4.0 * m * n * n * k if kind.startswith("attn") else 2.0 * m * n * k
    if count_cross_attn_qo:
        tot["linear"] += tot["cross_qo"]
    tot["total"] = sum(v for key, v in tot.items() if key != "cross_qo")
    return tot

"""ORACLE (test infrastructure, never shipped on the product path).

Plain-PyTorch fp32 restatement of the third-party UNet the reference calls at
``/root/reference/src/models/svd_unet.py:389,400,416``:
``diffusers.models.UNetSpatioTemporalConditionModel`` (diffusers is declared ``>=0.20.0`` in
``/root/reference/requirements.txt:2`` and was run at 0.36.0 per
``/root/reference/EXPERIMENT_REPORT.md:39``).  diffusers is NOT vendored in the reference and
not installed here, so this file restates its published architecture (SVD / SVD-XT config:
``block_out_channels (320,640,1280,1280)``, ``layers_per_block 2``, ``num_attention_heads
(5,10,20,20)``, ``cross_attention_dim 1024``, ``addition_time_embed_dim 256``,
``projection_class_embeddings_input_dim 768``, ``transformer_layers_per_block 1``).

**Parity unpinned**: the reference holds no golden vector, known-answer test or fixture at the
UNet boundary (``/root/reference/tests`` never touches ``svd_unet``), so this restatement is
anchored only on the reference's call site (argument names/shapes, ``svd_unet.py:416-422``)
and on the module/parameter naming of diffusers so that a real ``unet`` state_dict loads.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this.

Layout is the diffusers one (NCHW / (B,F,C,H,W)); the HIP engine uses NHWC and is compared
against this module on identical weights and inputs.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class SVDUNetConfig:
    in_channels: int = 8
    out_channels: int = 4
    block_out_channels: Sequence[int] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    num_attention_heads: Sequence[int] = (5, 10, 20, 20)
    cross_attention_dim: int = 1024
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 768
    norm_groups: int = 32
    # which down blocks carry transformers (SVD: all but the last)
    down_has_attn: Sequence[bool] = (True, True, True, False)

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @staticmethod
    def svd() -> "SVDUNetConfig":
        return SVDUNetConfig()

    @staticmethod
    def tiny(c: int = 64) -> "SVDUNetConfig":
        """Same topology, narrow channels (head dim stays 64) – for fast parity tests."""
        return SVDUNetConfig(
            block_out_channels=(c, 2 * c, 4 * c, 4 * c),
            num_attention_heads=(c // 64, 2 * c // 64, 4 * c // 64, 4 * c // 64),
            cross_attention_dim=128,
            addition_time_embed_dim=32,
            projection_class_embeddings_input_dim=96,
        )


def sinusoid(values: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers ``Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)``: [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32,
                                                       device=values.device) / half)
    args = values.float()[:, None] * freqs[None, :]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim: int, hidden: int, out_dim: int | None = None):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, hidden)
        self.linear_2 = nn.Linear(hidden, out_dim or hidden)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class AlphaBlender(nn.Module):
    """``learned_with_images`` with an all-video indicator: alpha = sigmoid(mix_factor)."""

    def __init__(self, alpha: float = 0.5):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.tensor([alpha]))

    def forward(self, x_spatial, x_temporal):
        a = torch.sigmoid(self.mix_factor).to(x_spatial.dtype)
        return a * x_spatial + (1.0 - a) * x_temporal


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb, eps, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class TemporalResnetBlock(nn.Module):
    def __init__(self, cin, cout, temb, eps, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv3d(cin, cout, (3, 1, 1), padding=(1, 0, 0))
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv3d(cout, cout, (3, 1, 1), padding=(1, 0, 0))
        self.conv_shortcut = nn.Conv3d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):  # x (B,C,F,H,W), temb (B,F,T)
        h = self.conv1(F.silu(self.norm1(x)))
        t = self.time_emb_proj(F.silu(temb))[:, :, :, None, None].permute(0, 2, 1, 3, 4)
        h = h + t
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class SpatioTemporalResBlock(nn.Module):
    def __init__(self, cin, cout, temb, eps, groups):
        super().__init__()
        self.spatial_res_block = ResnetBlock2D(cin, cout, temb, eps, groups)
        self.temporal_res_block = TemporalResnetBlock(cout, cout, temb, eps, groups)
        self.time_mixer = AlphaBlender(0.5)

    def forward(self, x, temb, num_frames):
        x = self.spatial_res_block(x, temb)
        bf, c, h, w = x.shape
        b = bf // num_frames
        xs = x.reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        xt = self.temporal_res_block(xs, temb.reshape(b, num_frames, -1))
        x = self.time_mixer(xs, xt)
        return x.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


class Attention(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim=None):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_k = nn.Linear(cross_dim or dim, inner, bias=False)
        self.to_v = nn.Linear(cross_dim or dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, dim), nn.Identity()])

    def forward(self, x, context=None):
        ctx = x if context is None else context
        b, n, _ = x.shape
        q = self.to_q(x).view(b, n, self.heads, -1).transpose(1, 2)
        k = self.to_k(ctx).view(b, ctx.shape[1], self.heads, -1).transpose(1, 2)
        v = self.to_v(ctx).view(b, ctx.shape[1], self.heads, -1).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v)
        o = o.transpose(1, 2).reshape(b, n, -1)
        return self.to_out[0](o)


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4):
        super().__init__()
        inner = dim * mult
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Identity(), nn.Linear(inner, dim_out or dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, dim_head, cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, context):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context) + x
        return self.ff(self.norm3(x)) + x


class TemporalBasicTransformerBlock(nn.Module):
    def __init__(self, dim, inner, heads, dim_head, cross_dim):
        super().__init__()
        self.is_res = dim == inner
        self.norm_in = nn.LayerNorm(dim)
        self.ff_in = FeedForward(dim, dim_out=inner)
        self.norm1 = nn.LayerNorm(inner)
        self.attn1 = Attention(inner, heads, dim_head)
        self.norm2 = nn.LayerNorm(inner)
        self.attn2 = Attention(inner, heads, dim_head, cross_dim)
        self.norm3 = nn.LayerNorm(inner)
        self.ff = FeedForward(inner)

    def forward(self, x, num_frames, context):
        bf, s, c = x.shape
        b = bf // num_frames
        x = x.reshape(b, num_frames, s, c).permute(0, 2, 1, 3).reshape(b * s, num_frames, c)
        res = x
        x = self.ff_in(self.norm_in(x))
        if self.is_res:
            x = x + res
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context) + x
        y = self.ff(self.norm3(x))
        x = y + x if self.is_res else y
        return x.reshape(b, s, num_frames, c).permute(0, 2, 1, 3).reshape(bf, s, c)


class TransformerSpatioTemporalModel(nn.Module):
    def __init__(self, heads, dim_head, channels, cross_dim, groups):
        super().__init__()
        inner = heads * dim_head
        self.channels = channels
        self.norm = nn.GroupNorm(groups, channels, eps=1e-6)
        self.proj_in = nn.Linear(channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_dim)])
        self.temporal_transformer_blocks = nn.ModuleList(
            [TemporalBasicTransformerBlock(inner, inner, heads, dim_head, cross_dim)])
        self.time_pos_embed = TimestepEmbedding(channels, channels * 4, out_dim=channels)
        self.time_mixer = AlphaBlender(0.5)
        self.proj_out = nn.Linear(inner, channels)

    def forward(self, x, context, num_frames):
        bf, c, h, w = x.shape
        b = bf // num_frames
        # temporal blocks attend to the FIRST frame's context, broadcast over all pixels
        tctx = context.reshape(b, num_frames, -1, context.shape[-1])[:, 0]
        tctx = tctx[:, None].expand(b, h * w, tctx.shape[-2], tctx.shape[-1])
        tctx = tctx.reshape(b * h * w, -1, tctx.shape[-1])

        res = x
        t = self.norm(x).permute(0, 2, 3, 1).reshape(bf, h * w, c)
        t = self.proj_in(t)
        frame_ids = torch.arange(num_frames, device=x.device).repeat(b)
        emb = self.time_pos_embed(sinusoid(frame_ids, self.channels).to(t.dtype))[:, None, :]
        for blk, tblk in zip(self.transformer_blocks, self.temporal_transformer_blocks):
            t = blk(t, context)
            tm = tblk(t + emb, num_frames, tctx)
            t = self.time_mixer(t, tm)
        t = self.proj_out(t)
        return t.reshape(bf, h, w, c).permute(0, 3, 1, 2) + res


class _Down(nn.Module):
    def __init__(self, cin, cout, temb, layers, heads, cross_dim, groups, attn, downsample):
        super().__init__()
        eps = 1e-6 if attn else 1e-5
        self.resnets = nn.ModuleList(
            [SpatioTemporalResBlock(cin if i == 0 else cout, cout, temb, eps, groups)
             for i in range(layers)])
        self.attentions = nn.ModuleList(
            [TransformerSpatioTemporalModel(heads, cout // heads, cout, cross_dim, groups)
             for _ in range(layers)]) if attn else None
        if downsample:
            ds = nn.Module()
            ds.conv = nn.Conv2d(cout, cout, 3, stride=2, padding=1)
            self.downsamplers = nn.ModuleList([ds])
        else:
            self.downsamplers = None

    def forward(self, x, temb, context, nf):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, temb, nf)
            if self.attentions is not None:
                x = self.attentions[i](x, context, nf)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0].conv(x)
            outs.append(x)
        return x, outs


class _Mid(nn.Module):
    def __init__(self, c, temb, heads, cross_dim, groups):
        super().__init__()
        self.resnets = nn.ModuleList(
            [SpatioTemporalResBlock(c, c, temb, 1e-5, groups) for _ in range(2)])
        self.attentions = nn.ModuleList(
            [TransformerSpatioTemporalModel(heads, c // heads, c, cross_dim, groups)])

    def forward(self, x, temb, context, nf):
        x = self.resnets[0](x, temb, nf)
        x = self.attentions[0](x, context, nf)
        return self.resnets[1](x, temb, nf)


class _Up(nn.Module):
    def __init__(self, cin, cout, prev, temb, layers, heads, cross_dim, groups, attn, upsample):
        super().__init__()
        rs = []
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            rin = prev if i == 0 else cout
            rs.append(SpatioTemporalResBlock(rin + skip, cout, temb, 1e-6, groups))
        self.resnets = nn.ModuleList(rs)
        self.attentions = nn.ModuleList(
            [TransformerSpatioTemporalModel(heads, cout // heads, cout, cross_dim, groups)
             for _ in range(layers)]) if attn else None
        if upsample:
            us = nn.Module()
            us.conv = nn.Conv2d(cout, cout, 3, padding=1)
            self.upsamplers = nn.ModuleList([us])
        else:
            self.upsamplers = None

    def forward(self, x, skips, temb, context, nf):
        for i, r in enumerate(self.resnets):
            x = torch.cat([x, skips.pop()], dim=1)
            x = r(x, temb, nf)
            if self.attentions is not None:
                x = self.attentions[i](x, context, nf)
        if self.upsamplers is not None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = self.upsamplers[0].conv(x)
        return x


class SVDUNetRef(nn.Module):
    """fp32 restatement of ``UNetSpatioTemporalConditionModel.forward`` (see module docstring)."""

    def __init__(self, cfg: SVDUNetConfig | None = None):
        super().__init__()
        cfg = cfg or SVDUNetConfig.svd()
        self.cfg = cfg
        boc = list(cfg.block_out_channels)
        temb = cfg.time_embed_dim
        g = cfg.norm_groups
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, temb)

        self.down_blocks = nn.ModuleList()
        ch = boc[0]
        for i, cout in enumerate(boc):
            last = i == len(boc) - 1
            self.down_blocks.append(_Down(ch, cout, temb, cfg.layers_per_block,
                                          cfg.num_attention_heads[i], cfg.cross_attention_dim, g,
                                          attn=cfg.down_has_attn[i], downsample=not last))
            ch = cout
        self.mid_block = _Mid(boc[-1], temb, cfg.num_attention_heads[-1],
                              cfg.cross_attention_dim, g)

        rev = boc[::-1]
        rev_heads = list(cfg.num_attention_heads)[::-1]
        rev_attn = list(cfg.down_has_attn)[::-1]
        self.up_blocks = nn.ModuleList()
        out_ch = rev[0]
        for i in range(len(rev)):
            prev = out_ch
            out_ch = rev[i]
            in_ch = rev[min(i + 1, len(rev) - 1)]
            self.up_blocks.append(_Up(in_ch, out_ch, prev, temb, cfg.layers_per_block + 1,
                                      rev_heads[i], cfg.cross_attention_dim, g,
                                      attn=rev_attn[i], upsample=i != len(rev) - 1))
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-5)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids, return_dict=False):
        cfg = self.cfg
        b, nf = sample.shape[:2]
        ts = torch.as_tensor(timestep, dtype=torch.float32, device=sample.device).reshape(-1)
        ts = ts.expand(b)
        emb = self.time_embedding(sinusoid(ts, cfg.block_out_channels[0]).to(sample.dtype))
        tids = sinusoid(added_time_ids.flatten(), cfg.addition_time_embed_dim)
        emb = emb + self.add_embedding(tids.reshape(b, -1).to(sample.dtype))

        x = sample.flatten(0, 1)
        emb = emb.repeat_interleave(nf, dim=0)
        ctx = encoder_hidden_states.repeat_interleave(nf, dim=0)

        x = self.conv_in(x)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, emb, ctx, nf)
            skips.extend(outs)
        x = self.mid_block(x, emb, ctx, nf)
        for blk in self.up_blocks:
            x = blk(x, skips, emb, ctx, nf)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        x = x.reshape(b, nf, *x.shape[1:])
        return (x,)


def unet_flops(cfg: SVDUNetConfig, frames: int, h: int, w: int, count_cross_attn_qo: bool = True):
    """Algorithmic FLOPs of one forward, by op class (2*M*N*K per contraction).

    Mirrors the op inventory in SURVEY.md section 8(d); used by bench.py for the MFMA roofline.
    """
    tot = {"conv3x3": 0.0, "tconv": 0.0, "conv1x1": 0.0, "linear": 0.0, "attn_s": 0.0,
           "attn_t": 0.0, "cross_qo": 0.0}
    temb = cfg.time_embed_dim

    def res(cin, cout, hh, ww):
        m = frames * hh * ww
        tot["conv3x3"] += 2 * m * 9 * cin * cout + 2 * m * 9 * cout * cout
        tot["tconv"] += 2 * 2 * m * 3 * cout * cout
        if cin != cout:
            tot["conv1x1"] += 2 * m * cin * cout
        tot["linear"] += 2 * 2 * temb * cout

    def xf(c, hh, ww):
        m = frames * hh * ww
        heads = c // 64
        tot["linear"] += 2 * m * c * c * 2            # proj_in / proj_out
        for _ in range(2):                            # spatial + temporal block
            tot["linear"] += 2 * m * c * c * 4        # q,k,v,o self-attn
            tot["cross_qo"] += 2 * m * c * c * 2      # cross-attn q + out (k,v on 1 token ~0)
            tot["linear"] += 2 * m * c * 8 * c + 2 * m * 4 * c * c   # GEGLU ff
        tot["linear"] += 2 * m * c * 8 * c + 2 * m * 4 * c * c       # ff_in
        tot["attn_s"] += 4 * frames * heads * (hh * ww) ** 2 * 64
        tot["attn_t"] += 4 * hh * ww * heads * frames ** 2 * 64

    boc = list(cfg.block_out_channels)
    hh, ww = h, w
    m0 = frames * h * w
    tot["conv3x3"] += 2 * m0 * 9 * cfg.in_channels * boc[0]
    ch = boc[0]
    for i, cout in enumerate(boc):
        for j in range(cfg.layers_per_block):
            res(ch if j == 0 else cout, cout, hh, ww)
            if cfg.down_has_attn[i]:
                xf(cout, hh, ww)
        ch = cout
        if i != len(boc) - 1:
            hh, ww = (hh + 1) // 2, (ww + 1) // 2
            tot["conv3x3"] += 2 * frames * hh * ww * 9 * cout * cout
    res(ch, ch, hh, ww); xf(ch, hh, ww); res(ch, ch, hh, ww)
    rev = boc[::-1]
    rev_attn = list(cfg.down_has_attn)[::-1]
    out_ch = rev[0]
    for i in range(len(rev)):
        prev, out_ch = out_ch, rev[i]
        in_ch = rev[min(i + 1, len(rev) - 1)]
        layers = cfg.layers_per_block + 1
        for j in range(layers):
            skip = in_ch if j == layers - 1 else out_ch
            rin = prev if j == 0 else out_ch
            res(rin + skip, out_ch, hh, ww)
            if rev_attn[i]:
                xf(out_ch, hh, ww)
        if i != len(rev) - 1:
            hh, ww = hh * 2, ww * 2
            tot["conv3x3"] += 2 * frames * hh * ww * 9 * out_ch * out_ch
    tot["conv3x3"] += 2 * m0 * 9 * boc[0] * cfg.out_channels
    if count_cross_attn_qo:
        tot["linear"] += tot["cross_qo"]
    total = sum(v for k, v in tot.items() if k != "cross_qo")
    tot["total"] = total
    return tot

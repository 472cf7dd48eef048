"""Temporal VAE decoder of Stable Video Diffusion on MI355X (the last stage's decode).

Replaces ``vae.decode(chunk, num_frames=...)`` / ``decode_latents`` of the reference's demo
(``/root/reference/scripts/generate_video_demo.py:154-195``, ``vae`` = diffusers ``AutoencoderKLTemporalDecoder``)
with the kernels of ``libsvdpipe_hip.so``: like ``unet_hip.py`` nothing here runs a PyTorch operator on activations.

Layout and fusions are the UNet engine's: activations are one fp16 token matrix ``[F*H*W][C]`` (channels-last, frame
major), so the 2-D convolutions, the (3,1,1) convolutions over frames and the per-frame attention tokens all address
the same rows; residual adds and the AlphaBlender mix live in GEMM epilogues
(``a*s + (1-a)*(s + conv2(..)) = s + (1-a)*conv2(..)`` with ``a = 1 - sigmoid(mix_factor)`` here, because this
decoder's blender has ``switch_spatial_to_temporal_mix=True``); nearest x2 upsampling is folded into the following
convolution's gather; ``1/scaling_factor`` is applied while the latent is packed into rows, and ``time_conv_out`` writes
straight into the video tensor the caller gets.

Mid-block attention (ONE head of width 512 over the H*W tokens of each frame; 2.4 TFLOP at 14 x 72 x 128): composed
from the implicit-GEMM kernel instead of a dedicated flash kernel -- per frame ``S = (Q K^T)/sqrt(C)`` (K rows are the
weight operand), an in-place row softmax (``sp_softmax_rows_f16``), ``V^T = W_v X^T`` (so that V^T is a K-contiguous
weight operand) and ``O = P V^T^T + b_v`` (rows of P sum to one, so the value bias moves behind the product).  The
score matrix of one frame (170 MB at 9,216 tokens) is the only large scratch and is reused frame after frame.
Since round 5 the logits stay fp32 into the softmax wherever a frame has a multiple of 256 tokens
(``sp_gemm_f32out_f16`` + ``sp_softmax_rows_f32``, the probabilities overwrite the front of each row's logits), so that
trained weights whose logits exceed fp16's range or precision are handled the way the reference's fp32 upcast handles them;
other token counts keep fp16 scores with a saturating softmax.

Precision: fp16 storage / fp32 accumulation throughout.  The reference upcasts this VAE to fp32
(``force_upcast``, ``generate_video_demo.py:171-175``) because fp16 activations of the trained decoder can overflow;
with the random weights available here that cannot be probed, see DESIGN.md section 7.
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Iterator, Sequence

import torch

from ..hip import ops
from .unet_hip import _Dense, _Norm, _f32


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
        return VAEDecoderConfig(block_out_channels=(c, 2 * c, 4 * c, 4 * c))


# ---------------------------------------------------------------------------------------------- parameters
def param_inventory(cfg: VAEDecoderConfig) -> Iterator[tuple]:
    """(name, shape, fan) of ``AutoencoderKLTemporalDecoder.decoder`` (diffusers naming, ``decoder.`` stripped);
    fan > 0: weight/bias of a layer with that fan-in, 0: norm scale, -1: norm bias, -2: AlphaBlender mix_factor."""

    def conv(p, cin, cout, k):
        yield p + ".weight", (cout, cin) + k, cin * math.prod(k)
        yield p + ".bias", (cout,), cin * math.prod(k)

    def lin(p, cin, cout):
        yield p + ".weight", (cout, cin), cin
        yield p + ".bias", (cout,), cin

    def norm(p, c):
        yield p + ".weight", (c,), 0
        yield p + ".bias", (c,), -1

    def res(p, cin, cout):
        s, t = p + ".spatial_res_block", p + ".temporal_res_block"
        yield from norm(s + ".norm1", cin)
        yield from conv(s + ".conv1", cin, cout, (3, 3))
        yield from norm(s + ".norm2", cout)
        yield from conv(s + ".conv2", cout, cout, (3, 3))
        if cin != cout:
            yield from conv(s + ".conv_shortcut", cin, cout, (1, 1))
        yield from norm(t + ".norm1", cout)
        yield from conv(t + ".conv1", cout, cout, (3, 1, 1))
        yield from norm(t + ".norm2", cout)
        yield from conv(t + ".conv2", cout, cout, (3, 1, 1))
        yield p + ".time_mixer.mix_factor", (1,), -2

    ch = list(cfg.block_out_channels)
    c = ch[-1]
    yield from conv("conv_in", cfg.latent_channels, c, (3, 3))
    yield from res("mid_block.resnets.0", c, c)
    a = "mid_block.attentions.0"
    yield from norm(a + ".group_norm", c)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        yield from lin(f"{a}.{n}", c, c)
    yield from res("mid_block.resnets.1", c, c)
    rev = list(reversed(ch))
    prev = rev[0]
    for i, co in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            yield from res(f"up_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
        if i != len(rev) - 1:
            yield from conv(f"up_blocks.{i}.upsamplers.0.conv", co, co, (3, 3))
        prev = co
    yield from norm("conv_norm_out", ch[0])
    yield from conv("conv_out", ch[0], cfg.out_channels, (3, 3))
    yield from conv("time_conv_out", cfg.out_channels, cfg.out_channels, (3, 1, 1))


def random_state_dict(cfg: VAEDecoderConfig, seed: int = 0, device="cpu", dtype=torch.float16) -> dict:
    """Random weights of the exact architecture (``nn.Conv*`` / ``nn.Linear`` default ranges, norm scale ~1)."""
    gen = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape, fan in param_inventory(cfg):
        if fan > 0:
            t = (torch.rand(shape, generator=gen, device=device) * 2 - 1) / math.sqrt(fan)
        elif fan == 0:
            t = 1.0 + 0.1 * (torch.rand(shape, generator=gen, device=device) - 0.5)
        elif fan == -1:
            t = 0.1 * (torch.rand(shape, generator=gen, device=device) - 0.5)
        else:
            t = torch.rand(shape, generator=gen, device=device) - 0.5
        sd[name] = t.to(dtype)
    return sd


def param_count(cfg: VAEDecoderConfig) -> int:
    return sum(math.prod(s) for _, s, _ in param_inventory(cfg))


def encoder_param_inventory(cfg: VAEDecoderConfig, in_channels: int = 3) -> Iterator[tuple]:
    """(name, shape, fan) of ``AutoencoderKLTemporalDecoder.encoder`` (keys ``encoder.*``) and ``quant_conv``."""

    def conv(p, cin, cout, k):
        yield p + ".weight", (cout, cin) + k, cin * math.prod(k)
        yield p + ".bias", (cout,), cin * math.prod(k)

    def norm(p, c):
        yield p + ".weight", (c,), 0
        yield p + ".bias", (c,), -1

    def res(p, cin, cout):
        yield from norm(p + ".norm1", cin)
        yield from conv(p + ".conv1", cin, cout, (3, 3))
        yield from norm(p + ".norm2", cout)
        yield from conv(p + ".conv2", cout, cout, (3, 3))
        if cin != cout:
            yield from conv(p + ".conv_shortcut", cin, cout, (1, 1))

    ch = list(cfg.block_out_channels)
    yield from conv("encoder.conv_in", in_channels, ch[0], (3, 3))
    prev = ch[0]
    for i, c in enumerate(ch):
        for j in range(cfg.layers_per_block):
            yield from res(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
        if i != len(ch) - 1:
            yield from conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", c, c, (3, 3))
        prev = c
    c = ch[-1]
    yield from res("encoder.mid_block.resnets.0", c, c)
    a = "encoder.mid_block.attentions.0"
    yield from norm(a + ".group_norm", c)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        yield a + f".{n}.weight", (c, c), c
        yield a + f".{n}.bias", (c,), c
    yield from res("encoder.mid_block.resnets.1", c, c)
    yield from norm("encoder.conv_norm_out", c)
    yield from conv("encoder.conv_out", c, 2 * cfg.latent_channels, (3, 3))
    yield from conv("quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, (1, 1))


def random_encoder_state_dict(cfg: VAEDecoderConfig, seed: int = 0, device="cpu", dtype=torch.float16) -> dict:
    gen = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape, fan in encoder_param_inventory(cfg):
        if fan > 0:
            t = (torch.rand(shape, generator=gen, device=device) * 2 - 1) / math.sqrt(fan)
        elif fan == 0:
            t = 1.0 + 0.1 * (torch.rand(shape, generator=gen, device=device) - 0.5)
        else:
            t = 0.1 * (torch.rand(shape, generator=gen, device=device) - 0.5)
        sd[name] = t.to(dtype)
    return sd


# ---------------------------------------------------------------------------------------------- engine
class _VAEKernels:
    """Kernel helpers shared by the decoder and the encoder engines."""

    def _init_common(self, cfg, device):
        self.cfg = cfg
        self.device = dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError(f"{type(self).__name__} runs on an MI355X HIP device only (no CPU fallback)")
        ops.load()  # fail loudly now if the extension is missing
        if any(v % 64 for v in cfg.block_out_channels):
            raise ValueError("block_out_channels must be multiples of 64 (MFMA K-steps)")
        self._ws = {}
        # mid-block attention logits in fp32 wherever the token count allows (a multiple of 256); VDPP_VAE_FP16_SCORES=1
        # selects the round-3 composition with fp16 scores everywhere (A/B, tests)
        self.fp32_scores = __import__("os").environ.get("VDPP_VAE_FP16_SCORES") != "1"
        self.gn_from_epilogue = __import__("os").environ.get("VDPP_GN_EPILOGUE", "1") != "0"
        return dev

    def _attn(self, sd, p, c):
        dev = self.device
        return dict(c=c, norm=_Norm(sd, p + ".group_norm", dev, 1e-6), q=_Dense.linear(sd, p + ".to_q", dev),
                    k=_Dense.linear(sd, p + ".to_k", dev),
                    wv=sd[p + ".to_v.weight"].to(dev, torch.float16).contiguous(),      # A operand of V^T = W_v X^T
                    bv=_f32(sd[p + ".to_v.bias"], dev), out=_Dense.linear(sd, p + ".to_out.0", dev))

    def _buf(self, rows, c):
        return torch.empty((rows, c), dtype=torch.float16, device=self.device)

    def _gemm(self, layer: _Dense, a, m, *, conv=None, temporal=None, gn_rows=0, **kw):
        """``gn_rows`` > 0: the output goes straight into a GroupNorm whose instances are multiples of that many rows (a
        frame): where the tiles allow it (256- / 320-wide column tiles, whole 256-row tiles per frame) the epilogue leaves
        per-tile column sums beside the output and ``_gn`` folds them instead of reading the tensor again (round 5,
        ``sp_gemm_desc.gn_part``; the 128-channel level has no such tiles)."""
        out = kw.pop("out", None)
        if out is None:
            out = self._buf(m, layer.n_true)
        n_store = layer.n_true if layer.n_true != layer.n else 0
        part = None
        if (gn_rows and self.gn_from_epilogue and gn_rows % 256 == 0 and m % 256 == 0 and n_store == 0
                and (layer.n % 256 == 0 or layer.n % 320 == 0)):
            part = torch.empty((m // 256, 2, layer.n, 2), dtype=torch.float32, device=self.device)
            kw.update(gn_part=part)
        ops.gemm(a, layer.w, out, m=m, n=layer.n, cin=layer.cin, mode=layer.mode, conv=conv, temporal=temporal,
                 bias=layer.bias, n_store=n_store, ldd=layer.n_true, lda=a.shape[1], **kw)
        if part is not None:
            out._gn_tile_sums = (part, out.data_ptr(), tuple(out.shape))
        return out

    def _gn(self, norm: _Norm, x, inst, rows, silu):
        c = x.shape[1]
        have = getattr(x, "_gn_tile_sums", None)
        if have is not None and have[1] == x.data_ptr() and have[2] == tuple(x.shape) and rows % 256 == 0:
            y = self._buf(x.shape[0], c)
            stats = torch.empty((inst, self.cfg.norm_groups, 2), dtype=torch.float32, device=self.device)
            ops.groupnorm_tile_sums(x, have[0], norm.g, norm.b, y, instances=inst, rows=rows, c=c, groups=self.cfg.norm_groups,
                                    eps=norm.eps, silu=silu, stats=stats)
            return y
        need = ops.groupnorm_ws_bytes(inst, rows, c, self.cfg.norm_groups)
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = self._ws[key] = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=self.device)
        y = self._buf(x.shape[0], c)
        ops.groupnorm(x, norm.g, norm.b, y, instances=inst, rows=rows, c=c, groups=self.cfg.norm_groups, eps=norm.eps,
                      silu=silu, ws=ws)
        return y

    def _run_attn(self, p, x, n_img, hw):
        """diffusers ``Attention(heads=1, dim_head=C, norm_num_groups=32, residual_connection=True)`` per image."""
        c, m = p["c"], n_img * hw
        if hw % 64:
            raise ValueError(f"mid-block attention: H*W = {hw} tokens per image must be a multiple of 64")
        t = self._gn(p["norm"], x, n_img, hw, False)
        q = self._gemm(p["q"], t, m)
        k = self._gemm(p["k"], t, m)
        o = self._buf(m, c)
        vt = self._buf(c, hw)
        scale = 1.0 / math.sqrt(c)
        if hw % 256 == 0 and c % 64 == 0 and self.fp32_scores:
            # Logits in fp32 all the way into the softmax (round 5; the reference runs this VAE in fp32 because trained
            # weights overflow fp16, generate_video_demo.py:171-175): Q K^T leaves the contraction as raw fp32 sums, the
            # softmax scales, normalises and writes the fp16 probabilities over the front of each row's logits, and the
            # second contraction reads them with twice the row pitch.  No logit is ever rounded to fp16 or clamped.
            scores32 = torch.empty((hw, hw), dtype=torch.float32, device=self.device)
            probs = scores32.view(torch.float16)                                          # [hw][2*hw] halves, same memory
            for i in range(n_img):
                r = slice(i * hw, (i + 1) * hw)
                ops.gemm_f32out(q[r], k[r], scores32, m=hw, n=hw, k=c)                    # S = Q K^T (fp32)
                ops.softmax_rows_f32(scores32, probs, rows=hw, cols=hw, scale=scale, ld=hw, ldo=2 * hw)
                ops.gemm(p["wv"], t[r], vt, m=c, n=hw, cin=c)                             # V^T = W_v X^T   [C][tokens]
                ops.gemm(probs[:, :hw], vt, o[r], m=hw, n=c, cin=hw, lda=2 * hw, bias=p["bv"])   # O = P V + b_v
            del scores32, probs
        else:                                   # token counts the fp32-output tiles do not cover: fp16 scores (saturating)
            scores = self._buf(hw, hw)
            for i in range(n_img):
                r = slice(i * hw, (i + 1) * hw)
                ops.gemm(q[r], k[r], scores, m=hw, n=hw, cin=c, oscale=scale)             # S = Q K^T / sqrt(C)
                ops.softmax_rows(scores, rows=hw, cols=hw)
                ops.gemm(p["wv"], t[r], vt, m=c, n=hw, cin=c)                             # V^T = W_v X^T   [C][tokens]
                ops.gemm(scores, vt, o[r], m=hw, n=c, cin=hw, bias=p["bv"])                # O = P V + b_v
            del scores
        del vt, q, k, t
        return self._gemm(p["out"], o, m, res1=x, r1scale=1.0, gn_rows=hw)


class TemporalDecoderHIP(_VAEKernels):
    """``AutoencoderKLTemporalDecoder.decode`` on a HIP device.  ``state_dict``: the ``decoder.*`` entries of a diffusers
    vae checkpoint with the prefix stripped (or ``random_state_dict``)."""

    def __init__(self, cfg: VAEDecoderConfig, state_dict: dict, device):
        dev = self._init_common(cfg, device)
        sd = state_dict
        ch = list(cfg.block_out_channels)
        c = ch[-1]
        self.conv_in = _Dense.conv3x3(sd, "conv_in", dev)
        self.mid = (self._res(sd, "mid_block.resnets.0", c, c), self._attn(sd, "mid_block.attentions.0", c),
                    self._res(sd, "mid_block.resnets.1", c, c))
        self.up = []
        rev = list(reversed(ch))
        prev = rev[0]
        for i, co in enumerate(rev):
            res = [self._res(sd, f"up_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
                   for j in range(cfg.layers_per_block + 1)]
            us = _Dense.conv3x3(sd, f"up_blocks.{i}.upsamplers.0.conv", dev) if i != len(rev) - 1 else None
            self.up.append((res, us))
            prev = co
        self.norm_out = _Norm(sd, "conv_norm_out", dev, 1e-6)
        # conv_out: 3 real output channels, stored as 8 columns (16-byte rows for frames_out's reads)
        w, b = sd["conv_out.weight"], sd["conv_out.bias"]
        w8 = torch.zeros((8,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
        b8 = torch.zeros(8, dtype=b.dtype, device=b.device)
        w8[:cfg.out_channels], b8[:cfg.out_channels] = w, b
        self.conv_out = _Dense.conv3x3({"c.weight": w8, "c.bias": b8}, "c", dev)
        if cfg.out_channels != 3:
            raise ValueError("time_conv_out kernel is written for 3 output channels")
        self.tco_w = _f32(sd["time_conv_out.weight"][:, :, :, 0, 0], dev)      # [out][in][tap]
        self.tco_b = _f32(sd["time_conv_out.bias"], dev)

    def _res(self, sd, p, cin, cout):
        dev = self.device
        s, t = p + ".spatial_res_block", p + ".temporal_res_block"
        sig = float(torch.sigmoid(sd[p + ".time_mixer.mix_factor"].float()).item())
        return dict(cin=cin, cout=cout, temporal_weight=sig,          # blend = (1 - sig)*spatial + sig*temporal
                    n1=_Norm(sd, s + ".norm1", dev, 1e-6), c1=_Dense.conv3x3(sd, s + ".conv1", dev),
                    n2=_Norm(sd, s + ".norm2", dev, 1e-6), c2=_Dense.conv3x3(sd, s + ".conv2", dev),
                    sc=_Dense.linear(sd, s + ".conv_shortcut", dev) if cin != cout else None,
                    tn1=_Norm(sd, t + ".norm1", dev, 1e-5), tc1=_Dense.tconv(sd, t + ".conv1", dev),
                    tn2=_Norm(sd, t + ".norm2", dev, 1e-5), tc2=_Dense.tconv(sd, t + ".conv2", dev))

    # ------------------------------------------------------------------ blocks
    def _run_res(self, p, x, b, f, h, w, gn_next=True):
        """SpatioTemporalResBlock(temb_channels=None): ResnetBlock2D per frame, TemporalResnetBlock over the frames,
        blended by the (switched) AlphaBlender."""
        hw, m = h * w, b * f * h * w
        geom = (b * f, h, w, h, w, 1, 0)
        t = self._gn(p["n1"], x, b * f, hw, True)
        t = self._gemm(p["c1"], t, m, conv=geom, gn_rows=hw)
        t = self._gn(p["n2"], t, b * f, hw, True)
        skip = x if p["sc"] is None else self._gemm(p["sc"], x, m)
        s = self._gemm(p["c2"], t, m, conv=geom, res1=skip, r1scale=1.0, gn_rows=hw)
        del t, skip
        t = self._gn(p["tn1"], s, b, f * hw, True)
        t = self._gemm(p["tc1"], t, m, temporal=(f, hw), gn_rows=hw)
        t = self._gn(p["tn2"], t, b, f * hw, True)
        # (1-sig)*s + sig*(s + conv2(t)) = s + sig*conv2(t)
        return self._gemm(p["tc2"], t, m, temporal=(f, hw), oscale=p["temporal_weight"], res1=s, r1scale=1.0,
                          gn_rows=hw if gn_next else 0)

    # ------------------------------------------------------------------ public
    def _decode_chunk(self, src, src_strides, dst, dst_strides, *, flat0, n, frames_per_item, batch, frames, h, w,
                      scale):
        """Entries flat0 .. flat0+n-1 of the flattened (batch, frame) list of ``src`` -> the same entries of ``dst``;
        the temporal layers treat them as ``batch`` items of ``frames`` frames (n = batch * frames)."""
        rows = self._buf(n * h * w, self.conv_in.cin)
        ops.vae_pack_latent(src, rows, scale=scale, flat0=flat0, n=n, frames_per_item=frames_per_item,
                            strides=src_strides, h=h, w=w, cpad=self.conv_in.cin)
        b, f = batch, frames
        m = n * h * w
        x = self._gemm(self.conv_in, rows, m, conv=(n, h, w, h, w, 1, 0), gn_rows=h * w)
        x = self._run_res(self.mid[0], x, b, f, h, w)
        x = self._run_attn(self.mid[1], x, n, h * w)
        x = self._run_res(self.mid[2], x, b, f, h, w)
        for res, us in self.up:
            for j, p in enumerate(res):
                # (a level's last resnet feeds the upsampling convolution: no norm behind it)
                x = self._run_res(p, x, b, f, h, w, gn_next=not (us is not None and j == len(res) - 1))
            if us is not None:
                x = self._gemm(us, x, n * 4 * h * w, conv=(n, h, w, 2 * h, 2 * w, 1, 1), gn_rows=4 * h * w)
                h, w = 2 * h, 2 * w
        x = self._gn(self.norm_out, x, n, h * w, True)
        x = self._gemm(self.conv_out, x, n * h * w, conv=(n, h, w, h, w, 1, 0))
        ops.vae_frames_out(x, self.tco_w, self.tco_b, dst, batch=b, frames=f, h=h, w=w, flat0=flat0,
                           frames_per_item=frames_per_item, strides=dst_strides)

    def decode(self, z, num_frames: int):
        """``vae.decode(z, num_frames).sample`` (ref generate_video_demo.py:181): z (B*F, 4, H, W) fp16, already divided
        by the scaling factor -> (B*F, 3, 8H, 8W) fp16."""
        if z.dim() != 4 or z.shape[1] != self.cfg.latent_channels or z.shape[0] % num_frames:
            raise ValueError(f"decode expects (B*F, {self.cfg.latent_channels}, H, W) with B*F divisible by num_frames; "
                             f"got {tuple(z.shape)}, num_frames={num_frames}")
        if z.dtype != torch.float16 or z.device != self.device or not z.is_contiguous():
            raise TypeError("decode expects a contiguous float16 tensor on this decoder's device")
        bf, c, h, w = z.shape
        out = torch.empty((bf, 3, 8 * h, 8 * w), dtype=torch.float16, device=self.device)
        hw, ohw = h * w, 64 * h * w
        self._decode_chunk(z, (num_frames * c * hw, hw, c * hw), out, (num_frames * 3 * ohw, ohw, 3 * ohw), flat0=0, n=bf,
                           frames_per_item=num_frames, batch=bf // num_frames, frames=num_frames, h=h, w=w, scale=1.0)
        return out

    def decode_latents(self, latents, num_frames: int, decode_chunk_size: int = 14, *, check_finite: bool = False):
        """``/root/reference/scripts/generate_video_demo.py:154-195``: latents (B, 4, F, H, W) fp16 -> frames
        (B, 3, F, 8H, 8W) fp32.  Division by ``scaling_factor`` first; the flattened (B, F) list is decoded
        ``decode_chunk_size`` entries per decoder call, each call ONE batch item of that many frames (the temporal
        layers see one chunk at a time), exactly as the reference does -- including chunks that straddle two videos
        when the chunk size does not divide F.
        ``check_finite``: this engine stores activations in fp16 where the reference upcasts the VAE to fp32
        (``force_upcast``, ref :171-175, because activations of the TRAINED decoder can leave fp16's range); with the flag
        the frames are checked (one device synchronisation) and a non-finite value raises ``FloatingPointError`` naming
        the first bad frame instead of being returned as a picture."""
        if latents.dim() != 5 or latents.shape[1] != self.cfg.latent_channels or latents.shape[2] != num_frames:
            raise ValueError(f"decode_latents expects (B, {self.cfg.latent_channels}, F, H, W) with F = num_frames; "
                             f"got {tuple(latents.shape)}")
        if decode_chunk_size <= 0:
            raise ValueError("decode_chunk_size must be positive")
        if latents.dtype != torch.float16 or latents.device != self.device or not latents.is_contiguous():
            raise TypeError("decode_latents expects a contiguous float16 tensor on this decoder's device")
        b, c, f, h, w = latents.shape
        out = torch.empty((b, 3, f, 8 * h, 8 * w), dtype=torch.float32, device=self.device)
        hw, ohw = h * w, 64 * h * w
        for i in range(0, b * f, decode_chunk_size):
            n = min(decode_chunk_size, b * f - i)
            self._decode_chunk(latents, (c * f * hw, f * hw, hw), out, (3 * f * ohw, f * ohw, ohw), flat0=i, n=n,
                               frames_per_item=f, batch=1, frames=n, h=h, w=w, scale=1.0 / self.cfg.scaling_factor)
        if check_finite:
            bad = (~torch.isfinite(out)).flatten(3).any(-1).any(1)                       # (B, F)
            if bool(bad.any()):
                bi, fi = [int(v) for v in bad.nonzero()[0]]
                raise FloatingPointError(f"decode_latents: non-finite values in video {bi}, frame {fi}: an activation left "
                                         f"fp16's range (the reference runs this VAE in fp32, force_upcast)")
        return out


class ImageEncoderHIP(_VAEKernels):
    """``vae.encode(image).latent_dist.mode()`` of the reference's ``encode_image``
    (``/root/reference/scripts/generate_video_demo.py:117-148``) on a HIP device: the 2-D encoder of
    ``AutoencoderKLTemporalDecoder`` + ``quant_conv``, mean half only.  ``state_dict``: the checkpoint's ``encoder.*``
    and ``quant_conv.*`` entries (or ``random_encoder_state_dict``).

    Downsample2D pads BOTTOM/RIGHT by one and strides by two without padding; the implicit-GEMM kernel's stride-2
    gather pads symmetrically, which for an even size is exactly "top/left only".  The engine therefore runs the whole
    encoder on the image mirrored in both axes with every 3x3 kernel mirrored as well (GroupNorm, the 1x1 shortcuts
    and the attention do not care about pixel order), and un-mirrors the 4-channel latent while writing it out: no
    padded copies, no extra kernel variant.  ``quant_conv`` (1x1 on 8 channels) is composed into ``conv_out`` at load
    time (fp32, rounded once), keeping only the four mean channels that ``mode()`` returns."""

    def __init__(self, cfg: VAEDecoderConfig, state_dict: dict, device):
        dev = self._init_common(cfg, device)
        sd = dict(state_dict)
        ch = list(cfg.block_out_channels)
        lc = cfg.latent_channels
        # compose quant_conv into conv_out (mean channels only), 8 stored columns
        wq = sd["quant_conv.weight"].float()[:lc, :, 0, 0]                                   # (lc, 2lc)
        wc, bc = sd["encoder.conv_out.weight"].float(), sd["encoder.conv_out.bias"].float()
        w8 = torch.zeros((8,) + tuple(wc.shape[1:]))
        b8 = torch.zeros(8)
        w8[:lc] = torch.einsum("oc,cikl->oikl", wq, wc)
        b8[:lc] = wq @ bc + sd["quant_conv.bias"].float()[:lc]
        sd["encoder.conv_out_q.weight"], sd["encoder.conv_out_q.bias"] = w8, b8
        for k in list(sd):                      # mirror every 3x3 kernel (see class docstring)
            if k.endswith(".weight") and sd[k].dim() == 4 and sd[k].shape[-1] == 3:
                sd[k] = sd[k].flip(-1, -2)
        e = "encoder"
        self.conv_in = _Dense.conv3x3(sd, e + ".conv_in", dev)
        self.down = []
        prev = ch[0]
        for i, c in enumerate(ch):
            res = [self._res2d(sd, f"{e}.down_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
                   for j in range(cfg.layers_per_block)]
            ds = _Dense.conv3x3(sd, f"{e}.down_blocks.{i}.downsamplers.0.conv", dev) if i != len(ch) - 1 else None
            self.down.append((res, ds))
            prev = c
        c = ch[-1]
        self.mid = (self._res2d(sd, e + ".mid_block.resnets.0", c, c), self._attn(sd, e + ".mid_block.attentions.0", c),
                    self._res2d(sd, e + ".mid_block.resnets.1", c, c))
        self.norm_out = _Norm(sd, e + ".conv_norm_out", dev, 1e-6)
        self.conv_out = _Dense.conv3x3(sd, e + ".conv_out_q", dev)
        self.levels = len(ch) - 1

    def _res2d(self, sd, p, cin, cout):
        dev = self.device
        return dict(n1=_Norm(sd, p + ".norm1", dev, 1e-6), c1=_Dense.conv3x3(sd, p + ".conv1", dev),
                    n2=_Norm(sd, p + ".norm2", dev, 1e-6), c2=_Dense.conv3x3(sd, p + ".conv2", dev),
                    sc=_Dense.linear(sd, p + ".conv_shortcut", dev) if cin != cout else None)

    def _run_res2d(self, p, x, n, h, w):
        m, geom = n * h * w, (n, h, w, h, w, 1, 0)
        t = self._gn(p["n1"], x, n, h * w, True)
        t = self._gemm(p["c1"], t, m, conv=geom)
        t = self._gn(p["n2"], t, n, h * w, True)
        skip = x if p["sc"] is None else self._gemm(p["sc"], x, m)
        return self._gemm(p["c2"], t, m, conv=geom, res1=skip, r1scale=1.0)

    def encode_image_latents(self, image, num_frames: int):
        """image: fp16 (B, 3, H, W) in [-1, 1], noise augmentation already added (ref :126-136).  Returns the
        ``image_latents`` of ``StableVideoUNet.set_conditioning``: fp16 (B, 4, num_frames, H/8, W/8), every frame the
        same latent (ref :139-148), no scaling factor."""
        if image.dim() != 4 or image.shape[1] != 3:
            raise ValueError(f"image must be (B, 3, H, W); got {tuple(image.shape)}")
        if image.dtype != torch.float16 or image.device != self.device or not image.is_contiguous():
            raise TypeError("image must be a contiguous float16 tensor on this encoder's device")
        n, _, h, w = image.shape
        f = 1 << self.levels
        if h % f or w % f or num_frames <= 0:
            raise ValueError(f"image height/width must be multiples of {f}, num_frames positive")
        if ((h // f) * (w // f)) % 64:
            raise ValueError("mid-block attention: (H/8)*(W/8) tokens must be a multiple of 64")
        rows = self._buf(n * h * w, self.conv_in.cin)
        ops.vae_image_pack(image, rows, batch=n, h=h, w=w, cpad=self.conv_in.cin, flip=True)
        x = self._gemm(self.conv_in, rows, n * h * w, conv=(n, h, w, h, w, 1, 0))
        for res, ds in self.down:
            for p in res:
                x = self._run_res2d(p, x, n, h, w)
            if ds is not None:
                x = self._gemm(ds, x, n * (h // 2) * (w // 2), conv=(n, h, w, h // 2, w // 2, 2, 0))
                h, w = h // 2, w // 2
        x = self._run_res2d(self.mid[0], x, n, h, w)
        x = self._run_attn(self.mid[1], x, n, h * w)
        x = self._run_res2d(self.mid[2], x, n, h, w)
        x = self._gn(self.norm_out, x, n, h * w, True)
        x = self._gemm(self.conv_out, x, n * h * w, conv=(n, h, w, h, w, 1, 0))
        out = torch.empty((n, self.cfg.latent_channels, num_frames, h, w), dtype=torch.float16, device=self.device)
        ops.vae_latent_out(x, out, batch=n, channels=self.cfg.latent_channels, frames=num_frames, h=h, w=w, flip=True)
        return out

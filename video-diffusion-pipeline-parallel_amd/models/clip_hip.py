"""CLIP image encoder of Stable Video Diffusion on MI355X (the first stage's encode).

Replaces ``image_encoder(pixel_values).image_embeds`` of the reference's demo
(``/root/reference/scripts/generate_video_demo.py:108-112``; ``image_encoder`` =
``transformers.CLIPVisionModelWithProjection`` loaded at ``:251-254``: ViT-H/14, 257 tokens of width 1280, 16 heads of 80,
32 pre-LayerNorm layers, projection to 1024) with the kernels of ``libsvdpipe_hip.so``.  Parameter names are the
transformers ones (``vision_model.*``, ``visual_projection.weight``), so the checkpoint's state_dict loads as it is.

One forward is 0.33 TFLOP once per video, so nothing here is tuned: the patch embedding is an im2col
(``sp_patchify_f16``) + the implicit-GEMM kernel with the position embeddings as a per-row bias, every LayerNorm in front
of a projection is folded into it (as in the UNet), attention over the 257 tokens is ``sp_attn_small_f16`` (K and V of a
head in LDS), the MLP activation is ``sp_gelu_f16``; residual adds live in GEMM epilogues.

Oracle: ``transformers`` itself is present in this image, so the parity test (``tests/test_clip_gpu.py``) compares with
the reference's own dependency on identical random weights -- the one third-party boundary of this repo that is pinned.
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from ..hip import ops
from . import weights as W
from .unet_hip import _Dense, _Norm, _f32


@dataclass
class CLIPVisionSpec:
    hidden_size: int = 1280
    intermediate_size: int = 5120
    num_hidden_layers: int = 32
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    projection_dim: int = 1024
    layer_norm_eps: float = 1e-5
    hidden_act: str = "gelu"

    @staticmethod
    def svd() -> "CLIPVisionSpec":
        """``image_encoder/config.json`` of stabilityai/stable-video-diffusion-img2vid(-xt): OpenCLIP ViT-H/14."""
        return CLIPVisionSpec()

    @staticmethod
    def from_config(cfg) -> "CLIPVisionSpec":
        """From a ``transformers.CLIPVisionConfig``."""
        return CLIPVisionSpec(cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.num_attention_heads,
                              cfg.image_size, cfg.patch_size, cfg.projection_dim, cfg.layer_norm_eps, cfg.hidden_act)


class CLIPVisionHIP:
    """``CLIPVisionModelWithProjection.forward(pixel_values).image_embeds`` on a HIP device."""

    def __init__(self, spec: CLIPVisionSpec, state_dict: dict, device):
        self.spec = spec
        self.device = dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("CLIPVisionHIP runs on an MI355X HIP device only (no CPU fallback)")
        ops.load()  # fail loudly now if the extension is missing
        c, heads = spec.hidden_size, spec.num_attention_heads
        if c % 64 or spec.intermediate_size % 64 or c % heads or (c // heads) % 8 or c // heads > 128:
            raise ValueError("unsupported widths: hidden/intermediate multiples of 64, head width a multiple of 8 <= 128")
        if spec.hidden_act not in ("gelu", "quick_gelu"):
            raise ValueError(f"hidden_act {spec.hidden_act!r} not supported")
        if spec.image_size % spec.patch_size:
            raise ValueError("image_size must be whole patches")
        sd = {k[len("vision_model."):] if k.startswith("vision_model.") else k: v for k, v in state_dict.items()}
        e = "embeddings"
        pw = sd[e + ".patch_embedding.weight"]                                  # (C, 3, P, P), no bias
        self.kpad = W.round_up(3 * spec.patch_size ** 2, 64)
        wp = torch.zeros(c, self.kpad, dtype=torch.float16, device=dev)
        wp[:, :3 * spec.patch_size ** 2] = pw.reshape(c, -1).to(dev, torch.float16)
        self.patch_w = wp
        pos = sd[e + ".position_embedding.weight"].to(dev).float()              # (1 + patches, C)
        self.n_patches = (spec.image_size // spec.patch_size) ** 2
        if pos.shape[0] != self.n_patches + 1:
            raise ValueError("position_embedding does not match image_size / patch_size")
        self.pos_patches = pos[1:].contiguous()                                 # per-row bias of the patch GEMM
        self.cls_row = (sd[e + ".class_embedding"].to(dev).float() + pos[0]).to(torch.float16).contiguous()
        self.pre_ln = _Norm(sd, "pre_layrnorm", dev, spec.layer_norm_eps)       # (sic: the transformers attribute name)
        self.layers = []
        for i in range(spec.num_hidden_layers):
            p = f"encoder.layers.{i}"
            a = p + ".self_attn"
            wqkv = torch.cat([sd[a + ".q_proj.weight"], sd[a + ".k_proj.weight"], sd[a + ".v_proj.weight"]], dim=0)
            bqkv = torch.cat([sd[a + ".q_proj.bias"], sd[a + ".k_proj.bias"], sd[a + ".v_proj.bias"]], dim=0)
            self.layers.append(dict(
                qkv=_Dense.fold_layernorm(wqkv, bqkv, sd[p + ".layer_norm1.weight"], sd[p + ".layer_norm1.bias"], dev,
                                          eps=spec.layer_norm_eps),
                out=_Dense.linear(sd, a + ".out_proj", dev),
                fc1=_Dense.fold_layernorm(sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"], sd[p + ".layer_norm2.weight"],
                                          sd[p + ".layer_norm2.bias"], dev, eps=spec.layer_norm_eps),
                fc2=_Dense.linear(sd, p + ".mlp.fc2", dev)))
        self.post_ln = _Norm(sd, "post_layernorm", dev, spec.layer_norm_eps)
        self.proj_w = W.pack_linear(state_dict["visual_projection.weight"]).to(dev)   # (projection_dim, C), no bias

    def _buf(self, rows, c):
        return torch.empty((rows, c), dtype=torch.float16, device=self.device)

    def _gemm(self, layer: _Dense, a, m, **kw):
        out = kw.pop("out", None)
        if out is None:
            out = self._buf(m, layer.n)
        ops.gemm(a, layer.w, out, m=m, n=layer.n, cin=layer.cin, bias=layer.bias, lda=a.shape[1],
                 ln_colsum=layer.colsum, **kw)
        return out

    def _ln_stats(self, layer: _Dense, x):
        st = torch.empty((x.shape[0], 2), dtype=torch.float32, device=self.device)
        ops.ln_stats(x, st, rows=x.shape[0], c=x.shape[1], eps=layer.ln_eps)
        return st

    def __call__(self, pixel_values):
        """pixel_values: fp16 (B, 3, image_size, image_size), already resized / normalised by the CLIPImageProcessor
        (ref generate_video_demo.py:108-109).  Returns ``image_embeds`` fp16 (B, projection_dim)."""
        sp = self.spec
        if (pixel_values.dim() != 4 or pixel_values.shape[1] != 3 or pixel_values.shape[2] != sp.image_size
                or pixel_values.shape[3] != sp.image_size):
            raise ValueError(f"pixel_values must be (B, 3, {sp.image_size}, {sp.image_size}); got {tuple(pixel_values.shape)}")
        if pixel_values.dtype != torch.float16 or pixel_values.device != self.device or not pixel_values.is_contiguous():
            raise TypeError("pixel_values must be a contiguous float16 tensor on this encoder's device")
        b, c, heads = pixel_values.shape[0], sp.hidden_size, sp.num_attention_heads
        npatch, seq, hd = self.n_patches, self.n_patches + 1, c // heads
        m = b * seq
        patches = self._buf(b * npatch, self.kpad)
        ops.patchify(pixel_values, patches, batch=b, h=sp.image_size, w=sp.image_size, patch=sp.patch_size, kpad=self.kpad)
        emb = self._buf(m, c)
        for i in range(b):      # [class token + pos 0 | patch embeddings + their position rows] per image
            emb[i * seq].copy_(self.cls_row)
            ops.gemm(patches[i * npatch:(i + 1) * npatch], self.patch_w, emb[i * seq + 1:(i + 1) * seq], m=npatch, n=c,
                     cin=self.kpad, bias2=self.pos_patches, bias2_rows=1)
        x = self._buf(m, c)
        ops.layernorm(emb, self.pre_ln.g, self.pre_ln.b, x, rows=m, c=c, eps=self.pre_ln.eps)
        scale = 1.0 / math.sqrt(hd)
        for L in self.layers:
            qkv = self._gemm(L["qkv"], x, m, ln_stats=self._ln_stats(L["qkv"], x))
            o = self._buf(m, c)
            ops.attn_small(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], o, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=b,
                           seq=seq, heads=heads, head_dim=hd, scale=scale)
            x = self._gemm(L["out"], o, m, res1=x, r1scale=1.0)
            f = self._gemm(L["fc1"], x, m, ln_stats=self._ln_stats(L["fc1"], x))
            ops.gelu(f, f, quick=sp.hidden_act == "quick_gelu")
            x = self._gemm(L["fc2"], f, m, res1=x, r1scale=1.0)
        # pooled = post_layernorm(last_hidden_state[:, 0]); image_embeds = visual_projection(pooled)
        pooled = self._buf(b, c)
        for i in range(b):
            ops.layernorm(x[i * seq:i * seq + 1], self.post_ln.g, self.post_ln.b, pooled[i:i + 1], rows=1, c=c,
                          eps=self.post_ln.eps)
        out = torch.empty((b, sp.projection_dim), dtype=torch.float16, device=self.device)
        ops.gemv(pooled, self.proj_w, None, n=sp.projection_dim, k=c, rows=b, y16=out)
        return out


def encode_image_embeddings(encoder: CLIPVisionHIP, pixel_values):
    """The CLIP half of the reference's ``encode_image`` (generate_video_demo.py:108-113):
    ``image_encoder(pixel_values).image_embeds.unsqueeze(1)`` -> (B, 1, projection_dim), the
    ``encoder_hidden_states`` that ``StableVideoUNet.set_conditioning`` takes."""
    return encoder(pixel_values).unsqueeze(1)

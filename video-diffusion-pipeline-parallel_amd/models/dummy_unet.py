"""``DummyUNet`` – the lightweight stand-in model of the simulator path.

Mirror of ``/root/reference/src/models/dummy_unet.py:18-59``:

    out = x + tanh(step / 10) * Conv3d(SiLU(Conv3d(x))) + LayerNorm_C(x)

with 3x3x3 / padding-1 convolutions ``C -> hidden -> C`` and a LayerNorm over the channel axis.
Parameter names (``net.0.*``, ``net.2.*``, ``norm.*``) are those of the reference so its
``state_dict`` loads unchanged.

Execution:
  * CPU tensors (Gloo simulator mode, BASELINE config 1): PyTorch CPU ops, bit-identical to the
    reference on the same seeds (pinned by ``tests/golden/dummy_*.npz``).
  * GPU tensors: one fused hand-written HIP kernel pair behind ``sp_dummy_unet_f32``
    (``csrc/dummy_unet.hip``).  There is no PyTorch-GPU fallback: a missing extension raises.
"""

from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn


class DummyUNet(nn.Module):
    def __init__(
        self,
        channels: int = 8,
        hidden_channels: int = 16,
        use_layernorm: bool = True,
    ) -> None:
        super().__init__()
        self.channels = channels
        self.hidden_channels = hidden_channels
        self.net = nn.Sequential(
            nn.Conv3d(channels, hidden_channels, kernel_size=3, padding=1),
            nn.SiLU(),
            nn.Conv3d(hidden_channels, channels, kernel_size=3, padding=1),
        )
        self.norm: Optional[nn.Module] = nn.LayerNorm(channels) if use_layernorm else None

    def forward(self, latent: torch.Tensor, step: int) -> torch.Tensor:  # type: ignore[override]
        gain = math.tanh(step / 10.0)
        if latent.is_cuda:
            return self._forward_hip(latent, gain)

        out = latent + gain * self.net(latent)
        if self.norm is not None:
            if latent.dim() < 2:
                raise ValueError("Latent tensor must have at least 2 dims (N, C, ...)")
            # LayerNorm over the channel axis: channels last, normalise, channels back.
            out = out + self.norm(latent.movedim(1, -1)).movedim(-1, 1)
        return out

    def _forward_hip(self, latent: torch.Tensor, gain: float) -> torch.Tensor:
        from ..hip import ops  # raises if libsvdpipe_hip.so is absent

        if latent.dim() != 5 or latent.dtype != torch.float32:
            raise ValueError("HIP DummyUNet path expects a (B, C, F, H, W) float32 latent")
        conv1, conv2 = self.net[0], self.net[2]
        return ops.dummy_unet_forward(
            latent.contiguous(),
            conv1.weight, conv1.bias, conv2.weight, conv2.bias,
            None if self.norm is None else self.norm.weight,
            None if self.norm is None else self.norm.bias,
            gain,
            1e-5 if self.norm is None else self.norm.eps,
        )

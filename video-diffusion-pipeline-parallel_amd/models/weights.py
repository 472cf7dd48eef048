"""Weight re-layout for the HIP kernels (done once at load time, on any device).

All dense weights are stored fp16 ``[N][K]`` with K contiguous, which is what the MFMA weight
operand wants (``nn.Linear.weight`` already is); convolution kernels are flattened tap-major so a
64-channel K-step never crosses a tap.
"""

from __future__ import annotations

import torch


def round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def pack_conv3x3(w: torch.Tensor, cin_pad: int | None = None, n_pad: int | None = None) -> torch.Tensor:
    """(Cout, Cin, 3, 3) -> (Npad, 9*Cpad) fp16 with k = (ky*3+kx)*Cpad + c."""
    cout, cin = w.shape[:2]
    cp = cin_pad or round_up(cin, 64)
    npad = n_pad or round_up(cout, 64)
    out = torch.zeros(npad, 9, cp, dtype=torch.float16, device=w.device)
    out[:cout, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, 9, cin).to(torch.float16)
    return out.reshape(npad, 9 * cp).contiguous()


def pack_tconv3(w: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin, 3, 1, 1) -> (Cout, 3*Cin) fp16 with k = tap*Cin + c."""
    cout, cin = w.shape[:2]
    return w[:, :, :, 0, 0].permute(0, 2, 1).reshape(cout, 3 * cin).to(torch.float16).contiguous()


def pack_linear(w: torch.Tensor) -> torch.Tensor:
    return w.reshape(w.shape[0], -1).to(torch.float16).contiguous()


def interleave_geglu(w: torch.Tensor, b: torch.Tensor | None):
    """GEGLU projection (2*inner, K): reorder rows so value/gate alternate in blocks of 16.

    Output rows [32q, 32q+16) are value rows 16q.., rows [32q+16, 32q+32) the matching gate rows, so a
    wave's adjacent 16-row MFMA sub-tiles hold (value, gate) for the same 16 output columns and the
    GEMM epilogue forms ``value * gelu(gate)`` in registers.
    """
    two_inner = w.shape[0]
    inner = two_inner // 2
    assert inner % 16 == 0
    idx = torch.arange(inner, device=w.device).reshape(-1, 16)
    order = torch.cat([idx, idx + inner], dim=1).reshape(-1)
    wi = w[order].to(torch.float16).contiguous()
    bi = None if b is None else b[order].float().contiguous()
    return wi, bi

"""Host-side Karras/Euler schedule tables used by ``StableVideoUNet``.

The reference obtains these from ``diffusers.EulerDiscreteScheduler`` (``/root/reference/src/models/
svd_unet.py:77-102``: scaled-linear betas, v-prediction, leading spacing, continuous timesteps,
``use_karras_sigmas=True``, ``sigma_min=0.002``, ``sigma_max=700``).  With explicit sigma_min/max and
Karras spacing the table has the closed form below; it is a few dozen floats computed once on the host.
"""

from __future__ import annotations

import torch

RHO = 7.0


def karras_sigma_table(num_steps: int, sigma_min: float = 0.002, sigma_max: float = 700.0) -> torch.Tensor:
    """float32 ``(num_steps + 1,)``; last entry 0."""
    ramp = torch.linspace(0.0, 1.0, num_steps, dtype=torch.float64)
    lo, hi = sigma_min ** (1.0 / RHO), sigma_max ** (1.0 / RHO)
    sig = (hi + ramp * (lo - hi)) ** RHO
    return torch.cat([sig, torch.zeros(1, dtype=torch.float64)]).to(torch.float32)


def continuous_timesteps(sigmas: torch.Tensor) -> torch.Tensor:
    """``0.25 * ln(sigma)`` for the non-terminal sigmas (continuous v-prediction conditioning)."""
    return 0.25 * torch.log(sigmas[:-1].to(torch.float32))

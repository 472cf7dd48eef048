"""ORACLE (test infrastructure): Karras-sigma Euler schedule used by the reference adapter.

Restates what ``/root/reference/src/models/svd_unet.py:77-102`` obtains from the third-party
``diffusers.EulerDiscreteScheduler`` (diffusers >=0.20.0 declared, 0.36.0 run; absent here):
``beta 0.00085->0.012 scaled_linear``, ``v_prediction``, ``timestep_spacing="leading"``,
``timestep_type="continuous"``, ``steps_offset=1``, ``use_karras_sigmas=True``,
``sigma_min=0.002``, ``sigma_max=700`` and then ``set_timesteps(N)``.

With explicit ``sigma_min/sigma_max`` and Karras spacing the published algorithm reduces to

    sigma_i = (smax^(1/7) + i/(N-1) * (smin^(1/7) - smax^(1/7)))^7   i = 0..N-1, then 0 appended
    t_i     = 0.25 * ln(sigma_i)                                      (continuous v-prediction)
    init_noise_sigma = sqrt(sigma_0^2 + 1)                            ("leading" spacing)

Pinned by the only values the reference documents: sigma_0 = 700.0 and init_noise_sigma ~ 700.0007
(``/root/reference/EXPERIMENT_RESULTS.md:237-251``).  Beyond that: parity unpinned.
"""

from __future__ import annotations

import numpy as np


def karras_sigmas(num_steps: int, sigma_min: float = 0.002, sigma_max: float = 700.0,
                  rho: float = 7.0) -> np.ndarray:
    """(N+1,) float32 sigma table, last entry 0."""
    ramp = np.linspace(0.0, 1.0, num_steps)
    lo, hi = sigma_min ** (1.0 / rho), sigma_max ** (1.0 / rho)
    sig = (hi + ramp * (lo - hi)) ** rho
    return np.concatenate([sig, [0.0]]).astype(np.float32)


def continuous_timesteps(sigmas: np.ndarray) -> np.ndarray:
    """(N,) float32: 0.25*log(sigma), evaluated in float32 like the torch code path."""
    return (np.float32(0.25) * np.log(sigmas[:-1].astype(np.float32))).astype(np.float32)


def init_noise_sigma(sigmas: np.ndarray) -> float:
    s0 = float(sigmas[0])
    return float((s0 * s0 + 1.0) ** 0.5)


def default_timestep_schedule(num_steps: int, num_train_timesteps: int = 1000) -> list[int]:
    """``StableVideoUNet._default_timestep_schedule`` (ref ``svd_unet.py:201-217``)."""
    ratio = num_train_timesteps // num_steps
    return list(range(num_train_timesteps - 1, -1, -ratio))[:num_steps]

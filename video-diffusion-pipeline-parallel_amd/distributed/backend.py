"""Choose the ``torch.distributed`` backend string.

Mirror of ``/root/reference/src/distributed/backend.py:12-31``.  On PyTorch-ROCm the string
``"nccl"`` selects RCCL (xGMI point-to-point between the GPUs of one node); ``"gloo"`` is the
CPU simulator transport.
"""

from __future__ import annotations

import os
from typing import Optional

SUPPORTED_BACKENDS = frozenset({"nccl", "gloo"})
BACKEND_ENV_VAR = "PIPELINE_BACKEND"


def resolve_backend(preferred: Optional[str] = None, *, simulator: bool = False) -> str:
    """Precedence: explicit argument > ``PIPELINE_BACKEND`` env > mode default.

    The mode default is ``"gloo"`` for the simulator and ``"nccl"`` (= RCCL) otherwise.
    An unsupported name from either source raises ``ValueError`` (ref ``backend.py:26-28``).
    """

    choice = (preferred or os.environ.get(BACKEND_ENV_VAR, "")).lower()
    if not choice:
        return "gloo" if simulator else "nccl"
    if choice not in SUPPORTED_BACKENDS:
        raise ValueError(f"Unsupported backend '{choice}'.")
    return choice

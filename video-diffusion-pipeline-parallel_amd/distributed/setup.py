"""Process-group bootstrap / teardown (mirror of ``/root/reference/src/distributed/setup.py:16-47``)."""

from __future__ import annotations

import logging
from datetime import timedelta
from typing import Optional

import torch.distributed as dist

LOGGER = logging.getLogger(__name__)

DEFAULT_TIMEOUT = timedelta(minutes=10)


def init_distributed(
    *,
    backend: str,
    rank: int,
    world_size: int,
    init_method: Optional[str] = None,
    timeout: Optional[timedelta] = None,
) -> None:
    """Create the default process group once; a second call is a no-op (ref ``setup.py:26-28``)."""

    if dist.is_initialized():
        LOGGER.debug("Process group already initialized.")
        return

    LOGGER.info(
        "Initializing process group backend=%s rank=%s world_size=%s", backend, rank, world_size
    )
    extra = {"init_method": init_method} if init_method else {}
    dist.init_process_group(
        backend=backend,
        rank=rank,
        world_size=world_size,
        timeout=timeout or DEFAULT_TIMEOUT,
        **extra,
    )


def finalize_distributed() -> None:
    """Destroy the default process group if one exists (ref ``setup.py:45-47``)."""

    if dist.is_initialized():
        dist.destroy_process_group()

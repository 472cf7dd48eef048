from .backend import resolve_backend
from .setup import finalize_distributed, init_distributed

__all__ = ["resolve_backend", "init_distributed", "finalize_distributed"]

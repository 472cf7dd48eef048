"""Models behind the pipeline's ``model(latent, step)`` protocol (ref ``src/models/__init__.py``)."""

from .dummy_unet import DummyUNet

__all__ = ["DummyUNet", "StableVideoUNet"]


def __getattr__(name):  # lazy: svd_unet pulls in the HIP binding
    if name == "StableVideoUNet":
        from .svd_unet import StableVideoUNet
        return StableVideoUNet
    raise AttributeError(name)

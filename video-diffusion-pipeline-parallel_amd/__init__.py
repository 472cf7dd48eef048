"""MI355X-native diffusion-step pipeline backend for Stable Video Diffusion.

Sub-packages mirror the reference layout (``/root/reference/src``) so the backend is a
drop-in for that project's pipeline path:

* ``pipeline``    – step assignment + per-rank stage executor (ref ``src/pipeline``)
* ``distributed`` – backend selection / process-group bootstrap (ref ``src/distributed``)
* ``models``      – ``DummyUNet`` and ``StableVideoUNet`` (ref ``src/models``); the SVD UNet
  forward is executed by hand-written gfx950 HIP kernels behind the C ABI in
  ``include/svdpipe.h`` (``csrc/``), never by a PyTorch/CPU fallback.
* ``modes``       – simulator / production / benchmark entrypoints (ref ``src/modes``)
* ``hip``         – ctypes binding of ``libsvdpipe_hip.so``
"""

__version__ = "0.1.0"

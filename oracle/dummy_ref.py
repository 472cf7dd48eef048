"""ORACLE (test infrastructure): ctypes front-end of ``oracle/dummy_unet_ref.c``."""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libdummy_ref.so")
    src = os.path.join(_HERE, "dummy_unet_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libdummy_ref.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def run_steps(latent: np.ndarray, timesteps, start: int, end: int, params: dict,
              ln_eps: float = 1e-5) -> np.ndarray:
    """``latent`` (B,C,F,H,W) fp32 -> latent after ``timesteps[start:end]``."""
    x = np.ascontiguousarray(latent, dtype=np.float32).copy()
    B, C, F, H, W = x.shape
    w1 = np.ascontiguousarray(params["net.0.weight"], np.float32)
    b1 = np.ascontiguousarray(params["net.0.bias"], np.float32)
    w2 = np.ascontiguousarray(params["net.2.weight"], np.float32)
    b2 = np.ascontiguousarray(params["net.2.bias"], np.float32)
    use_ln = "norm.weight" in params
    lw = np.ascontiguousarray(params["norm.weight"], np.float32) if use_ln else np.zeros(C, np.float32)
    lb = np.ascontiguousarray(params["norm.bias"], np.float32) if use_ln else np.zeros(C, np.float32)
    ts = np.ascontiguousarray(np.asarray(timesteps, dtype=np.int32))
    _lib().dummy_pipeline_run_ref(
        _p(x), _p(ts), ctypes.c_int(start), ctypes.c_int(end), _p(w1), _p(b1), _p(w2), _p(b2),
        _p(lw), _p(lb), ctypes.c_float(ln_eps), ctypes.c_int(int(use_ln)),
        *(ctypes.c_int(v) for v in (B, C, w1.shape[0], F, H, W)))
    return x

"""ctypes binding of ``libsvdpipe_hip.so`` (C ABI declared in ``include/svdpipe.h``).

The shared object is built in-tree by ``csrc/Makefile`` (``__graft_entry__.build()``).  There is
no fallback: importing :mod:`ops` without the library raises ``RuntimeError``.
"""

from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsvdpipe_hip.so")
_lib = None


class GemmDesc(ctypes.Structure):
    """Mirror of ``struct sp_gemm_desc`` (include/svdpipe.h)."""

    _fields_ = [
        ("a", ctypes.c_void_p), ("lda", ctypes.c_int64), ("mode", ctypes.c_int), ("cin", ctypes.c_int),
        ("n_img", ctypes.c_int), ("hin", ctypes.c_int), ("win", ctypes.c_int), ("hout", ctypes.c_int),
        ("wout", ctypes.c_int), ("stride", ctypes.c_int), ("upsample2x", ctypes.c_int),
        ("frames", ctypes.c_int), ("hw", ctypes.c_int64),
        ("w", ctypes.c_void_p), ("m", ctypes.c_int), ("n", ctypes.c_int),
        ("bias", ctypes.c_void_p), ("bias2", ctypes.c_void_p), ("bias2_rows", ctypes.c_int64),
        ("ldb2", ctypes.c_int64),
        ("res1", ctypes.c_void_p), ("ldr1", ctypes.c_int64), ("r1scale", ctypes.c_float),
        ("res2", ctypes.c_void_p), ("ldr2", ctypes.c_int64), ("r2scale", ctypes.c_float),
        ("oscale", ctypes.c_float), ("geglu", ctypes.c_int), ("n_store", ctypes.c_int),
        ("d", ctypes.c_void_p), ("ldd", ctypes.c_int64), ("zero_page", ctypes.c_void_p),
        ("ln_stats", ctypes.c_void_p), ("ln_colsum", ctypes.c_void_p),
        ("euler_latent", ctypes.c_void_p), ("euler_out", ctypes.c_void_p),
        ("euler_eps_uncond", ctypes.c_void_p), ("euler_ld_eps", ctypes.c_int64),
        ("euler_guidance", ctypes.c_void_p),
        ("euler_sigma", ctypes.c_float), ("euler_sigma_next", ctypes.c_float),
        ("euler_frames", ctypes.c_int), ("euler_hw", ctypes.c_int64),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
        ("ln_out", ctypes.c_void_p), ("ln_out_eps", ctypes.c_float),
        ("w_group_rows", ctypes.c_int64), ("w_group_stride", ctypes.c_int64),
        ("gn_part", ctypes.c_void_p),
        ("a2", ctypes.c_void_p), ("lda2", ctypes.c_int64), ("cin2", ctypes.c_int),
    ]


# name -> (restype, argtypes); every symbol declared in include/svdpipe.h
_P, _I, _L, _F, _Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
SIGNATURES = {
    "sp_last_error": (ctypes.c_char_p, []),
    "sp_version": (_I, []),
    "sp_gemm_f16": (_I, [ctypes.POINTER(GemmDesc), _P]),
    "sp_gemm_desc_size": (_Z, []),
    "sp_gemm_workspace_bytes": (_Z, [ctypes.POINTER(GemmDesc)]),
    "sp_gemm_set_route": (_I, [_I, _I, _I]),
    "sp_gemm_last_kernel": (ctypes.c_char_p, []),
    "sp_gemv_f16": (_I, [_P, _L, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "sp_gemv_batched_f16": (_I, [_P, _L, _L, _P, _L, _P, _L, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _P]),
    "sp_sinusoid_f16": (_I, [_P, _P, _I, _I, _P]),
    "sp_groupnorm_ws_bytes": (_Z, [_I, _L, _I, _I]),
    "sp_groupnorm_f16": (_I, [_P, _P, _P, _P, _I, _L, _I, _I, _F, _I, _P, _Z, _P]),
    "sp_groupnorm_ld_f16": (_I, [_P, _L, _P, _P, _P, _I, _L, _I, _I, _F, _I, _P, _Z, _P]),
    "sp_groupnorm_tile_sums_f16": (_I, [_P, _L, _P, _P, _P, _P, _I, _L, _I, _I, _F, _I, _P, _P]),
    "sp_groupnorm_tile_sums2_f16": (_I, [_P, _L, _P, _I, _P, _P, _P, _P, _I, _L, _I, _I, _F, _I, _P, _P]),
    "sp_groupnorm_fold_linear_tile_sums_f16": (_I, [_P, _P, _P, _I, _L, _I, _I, _F, _P, _P, _I, _P, _P, _P, _P]),
    "sp_groupnorm_fold_linear_f16": (_I, [_P, _L, _P, _P, _I, _L, _I, _I, _F, _P, _P, _I, _P, _P, _P, _Z, _P]),
    "sp_layernorm_f16": (_I, [_P, _P, _L, _P, _P, _P, _P, _L, _I, _F, _P]),
    "sp_ln_stats_f16": (_I, [_P, _P, _L, _P, _P, _L, _I, _F, _P]),
    "sp_attn_spatial_f16": (_I, [_P, _P, _P, _P, _L, _L, _L, _L, _I, _I, _I, _F, _P, _P]),
    "sp_attn_long_ws_bytes": (_L, [_I, _I, _I]),
    "sp_attn_spatial_long_f16": (_I, [_P, _P, _P, _P, _L, _L, _L, _L, _I, _I, _I, _F, _P, _P, _L, _P]),
    "sp_attn_fp8_ws_bytes": (_L, [_I, _I, _I]),
    "sp_attn_spatial_fp8": (_I, [_P, _P, _P, _P, _L, _L, _L, _L, _I, _I, _I, _F, _P, _L, _P, _P]),
    "sp_attn_temporal_f16": (_I, [_P, _P, _P, _P, _L, _L, _L, _L, _I, _I, _L, _I, _F, _P, _P]),
    "sp_pack_input_f16": (_I, [_P, _P, _P, _F, _I, _I, _I, _I, _I, _P]),
    "sp_euler_step_f16": (_I, [_P, _P, _P, _L, _P, _P, _F, _F, _I, _I, _I, _I, _P]),
    "sp_concat_channels_f16": (_I, [_P, _I, _P, _I, _P, _L, _P]),
    "sp_add_rowvec_f16": (_I, [_P, _P, _P, _L, _I, _P]),
    "sp_softmax_rows_f16": (_I, [_P, _L, _L, _I, _P]),
    "sp_gemm_f32out_f16": (_I, [_P, _L, _P, _P, _I, _I, _I, _P, _P]),
    "sp_softmax_rows_f32": (_I, [_P, _L, _P, _L, _L, _I, _F, _P]),
    "sp_vae_pack_latent_f16": (_I, [_P, _P, _F, _L, _I, _I, _L, _L, _L, _I, _I, _I, _P]),
    "sp_vae_frames_out_f16": (_I, [_P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _L, _I, _L, _L, _L, _P]),
    "sp_vae_image_pack_f16": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "sp_vae_latent_out_f16": (_I, [_P, _L, _P, _I, _I, _I, _I, _I, _I, _P]),
    "sp_patchify_f16": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "sp_attn_small_f16": (_I, [_P, _P, _P, _P, _L, _L, _L, _L, _I, _I, _I, _I, _F, _P]),
    "sp_gelu_f16": (_I, [_P, _P, _L, _I, _P]),
    "sp_clock_stamp": (_I, [_P, _I, _P]),
    "sp_dummy_unet_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _F, _I, _I, _I, _I, _I, _I, _P]),
}


def load():
    """Load the library once and attach prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is not built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C csrc`). "
                "There is no CPU/PyTorch fallback for the GPU path."
            )
        import torch  # noqa: F401  load PyTorch's HIP runtime first so both share one libamdhip64 instance

        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if lib.sp_gemm_desc_size() != ctypes.sizeof(GemmDesc):
            raise RuntimeError(f"GemmDesc mirrors {ctypes.sizeof(GemmDesc)} bytes, the library's sp_gemm_desc has "
                               f"{lib.sp_gemm_desc_size()}: rebuild the library or update hip/__init__.py")
        _lib = lib
    return _lib


def last_error() -> str:
    return load().sp_last_error().decode()

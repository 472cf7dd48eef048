"""Tensor-level wrappers over the C ABI: raw ``data_ptr()`` + the current HIP stream go in,
kernels are enqueued, nothing is synchronised.  PyTorch only owns the memory."""

from __future__ import annotations

import ctypes

import torch

from . import GemmDesc, last_error, load

A_LINEAR, A_CONV3X3, A_TEMPORAL3 = 0, 1, 2
ZERO_PAGE_BYTES = 4096
_zero_pages: dict = {}


class HipKernelError(RuntimeError):
    pass


# Optional per-launch timing (bench.py roofline leg): when a list is installed here, the contraction
# and attention wrappers bracket their launch with events on the launching stream and append
# (kind, flops, start_event, end_event, algorithmic_bytes, tag) -- tag: the shape of a contraction, else None.
PROFILE = None


class _Timed:
    def __init__(self, kind, flops, nbytes=0.0, tag=None):
        self.kind, self.flops, self.nbytes, self.tag = kind, flops, nbytes, tag

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.e1.record()
            PROFILE.append((self.kind, self.flops, self.e0, self.e1, self.nbytes, self.tag))
        return False


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise HipKernelError(f"{what} failed (rc={rc}): {last_error()}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def zero_page(device) -> torch.Tensor:
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _zero_pages:
        _zero_pages[key] = torch.zeros(ZERO_PAGE_BYTES, dtype=torch.uint8, device=dev)
    return _zero_pages[key]


def _f16(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float16 or not t.is_cuda:
        raise TypeError(f"{name} must be a float16 tensor on a HIP device")
    if t.dim() >= 1 and t.stride(-1) != 1 and t.shape[-1] != 1:
        raise ValueError(f"{name} must be row-major (innermost stride 1); got strides {tuple(t.stride())}")
    return t


def _rows(t: torch.Tensor, name: str, ld: int) -> torch.Tensor:
    """2-D token matrix whose row pitch must equal the leading dimension handed to the kernel."""
    _f16(t, name)
    if t.dim() == 2 and t.shape[0] > 1 and t.stride(0) != ld:
        raise ValueError(f"{name}: row stride {t.stride(0)} does not match leading dimension {ld}")
    return t


def gemm(a, w, out, *, m, n, cin, mode=A_LINEAR, lda=None, conv=None, temporal=None, bias=None,
         bias2=None, bias2_rows=0, ldb2=0, res1=None, r1scale=1.0, res2=None, r2scale=1.0, oscale=1.0,
         geglu=False, n_store=0, ldd=None, ldr1=None, ldr2=None, ln_stats=None, ln_colsum=None, euler=None,
         workspace=None, ln_out=None, ln_out_eps=1e-5, w_group_rows=0, w_group_stride=0, gn_part=None, a2=None, cin2=0,
         lda2=None):
    """``out[m][:] = epilogue(sum_taps A_tap @ W^T)``; see ``sp_gemm_desc`` in include/svdpipe.h.
    ``ln_stats`` / ``ln_colsum``: LayerNorm folded into the contraction (``a`` is the UN-normalised tensor).
    ``w_group_rows`` / ``w_group_stride``: ``w`` holds one weight matrix per group of that many output rows (a GroupNorm
    folded into this linear layer, ``groupnorm_fold_linear``)."""
    d = GemmDesc()
    d.lda = int(lda if lda is not None else cin)
    d.a, d.mode, d.cin = _rows(a, "a", d.lda).data_ptr(), mode, cin
    if mode == A_CONV3X3:
        d.n_img, d.hin, d.win, d.hout, d.wout, d.stride, d.upsample2x = conv
    if mode == A_TEMPORAL3:
        d.frames, d.hw = temporal
    d.w, d.m, d.n = _f16(w, "w").data_ptr(), m, n
    d.bias, d.bias2, d.bias2_rows, d.ldb2 = _ptr(bias), _ptr(bias2), bias2_rows, ldb2
    nout = n // 2 if geglu else n
    d.res1, d.ldr1, d.r1scale = _ptr(res1), int(ldr1 if ldr1 is not None else nout), r1scale
    d.res2, d.ldr2, d.r2scale = _ptr(res2), int(ldr2 if ldr2 is not None else nout), r2scale
    d.oscale, d.geglu, d.n_store = oscale, int(geglu), n_store
    d.ldd = int(ldd if ldd is not None else (n_store or nout))
    d.d = _rows(out, "out", d.ldd).data_ptr()
    d.zero_page = zero_page(a.device).data_ptr()
    d.ln_stats, d.ln_colsum = _ptr(ln_stats), _ptr(ln_colsum)
    if euler is not None:       # conv_out only: guidance mix + Euler update instead of storing eps rows (see svdpipe.h)
        d.euler_latent, d.euler_out = _f16(euler["latent"], "latent").data_ptr(), _f16(euler["out"], "out").data_ptr()
        d.euler_eps_uncond, d.euler_guidance = _ptr(euler.get("eps_uncond")), _ptr(euler.get("guidance"))
        d.euler_ld_eps = int(euler.get("ld_eps", 0))
        d.euler_sigma, d.euler_sigma_next = float(euler["sigma"]), float(euler["sigma_next"])
        d.euler_frames, d.euler_hw = int(euler["frames"]), int(euler["hw"])
    if workspace is not None:      # fp32 scratch: split-K (few rows, long K; too small a buffer simply disables it), or the
        # per-tile row sums of a two-tile ln_out
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    if ln_out is not None:         # (mean, rstd) of the output rows, for the next contraction's folded LayerNorm
        if ln_out.dtype != torch.float32 or tuple(ln_out.shape) != (m, 2) or not ln_out.is_contiguous():
            raise ValueError("ln_out must be a contiguous float32 [m][2] tensor")
        d.ln_out, d.ln_out_eps = ln_out.data_ptr(), float(ln_out_eps)
    d.w_group_rows, d.w_group_stride = int(w_group_rows), int(w_group_stride)
    if a2 is not None:                 # extra linear tap: rows of a second tensor against the weight's last cin2 columns
        _f16(a2, "a2")
        d.a2, d.lda2, d.cin2 = a2.data_ptr(), int(lda2 if lda2 is not None else a2.stride(0)), int(cin2)
    if gn_part is not None:            # fp32 [m/256][2][n][2]: per-tile column sums for the next GroupNorm (svdpipe.h)
        if gn_part.dtype != torch.float32 or not gn_part.is_cuda or gn_part.numel() < (m // 256) * 2 * n * 2:
            raise TypeError("gn_part must be a float32 HIP tensor of (m/256)*2*n*2 elements")
        d.gn_part = gn_part.data_ptr()
    taps = 9 if mode == A_CONV3X3 else 3 if mode == A_TEMPORAL3 else 1
    # algorithmic bytes of this launch: every operand element once (A without tap re-reads), output and residuals once
    a_rows = d.n_img * d.hin * d.win if mode == A_CONV3X3 else m
    k2 = int(cin2) if a2 is not None else 0          # (the extra linear tap's channels count like any other K)
    nbytes = 2.0 * (a_rows * cin + m * k2 + n * (taps * cin + k2) + m * (n_store or nout) * (1 + (res1 is not None) + (res2 is not None)))
    with _Timed("gemm", 2.0 * m * n * (taps * cin + k2), nbytes, (m, n, cin, mode, bool(geglu))) as tm:
        _check(load().sp_gemm_f16(ctypes.byref(d), _stream()), "sp_gemm_f16")
        if PROFILE is not None:      # which kernel template took it (per-template FLOPs in the profile summary)
            tm.tag = tm.tag + (load().sp_gemm_last_kernel().decode(),)
    return out


def gemm_workspace_bytes(*, m, n, cin, mode=A_LINEAR) -> int:
    """Scratch bytes ``gemm(..., workspace=)`` can use for this shape (0: not a split-K candidate)."""
    d = GemmDesc()
    d.m, d.n, d.cin, d.mode = m, n, cin, mode
    return int(load().sp_gemm_workspace_bytes(ctypes.byref(d)))


class gemm_route:
    """``with gemm_route(3, bm=256): ...`` pins the kernel family of ``sp_gemm_f16`` (tests / micro-benchmarks only;
    see ``sp_gemm_set_route`` in include/svdpipe.h): 1 small tiles, 2 ping-pong, 3 persistent-stream."""

    def __init__(self, route: int, bm: int = 0, bn: int = 0):
        self.args = (route, bm, bn)

    def __enter__(self):
        _check(load().sp_gemm_set_route(*self.args), "sp_gemm_set_route")
        return self

    def __exit__(self, *exc):
        load().sp_gemm_set_route(0, 0, 0)
        return False


def gemv(x, w, b, *, n, k, rows=1, ldx=None, y32=None, y16=None, ldy=None, silu_in=False, silu_out=False):
    _check(load().sp_gemv_f16(_f16(x, "x").data_ptr(), int(ldx if ldx is not None else k), _f16(w, "w").data_ptr(),
                              _ptr(b), _ptr(y32), _ptr(y16), int(ldy if ldy is not None else n), rows, n, k,
                              int(silu_in), int(silu_out), _stream()), "sp_gemv_f16")


def gemv_batched(x, w, b, *, batch, n, k, rows=1, ldx=None, x_stride=None, y32=None, y16=None, ldy=None,
                 silu_in=False, silu_out=False):
    """``batch`` same-shape GEMVs in one launch: w [batch][n][k], b [batch][n] or None, y [batch][rows][n];
    x [batch][rows][k], or one shared [rows][k] input with ``x_stride=0``."""
    ldx = int(ldx if ldx is not None else k)
    ldy = int(ldy if ldy is not None else n)
    xs = int(rows * ldx if x_stride is None else x_stride)
    _check(load().sp_gemv_batched_f16(_f16(x, "x").data_ptr(), ldx, xs, _f16(w, "w").data_ptr(), n * k, _ptr(b), n,
                                      _ptr(y32), _ptr(y16), ldy, rows * ldy, batch, rows, n, k, int(silu_in),
                                      int(silu_out), _stream()), "sp_gemv_batched_f16")


def sinusoid(values32, out16, count, dim):
    _check(load().sp_sinusoid_f16(values32.data_ptr(), out16.data_ptr(), count, dim, _stream()), "sp_sinusoid_f16")


def groupnorm_ws_bytes(instances, rows, c, groups) -> int:
    return int(load().sp_groupnorm_ws_bytes(instances, rows, c, groups))


def groupnorm(x, gamma, beta, y, *, instances, rows, c, groups, eps, silu, ws, ldx=None):
    """``ldx``: halves between consecutive rows of ``x`` when it is a column slice of a wider tensor (default: dense)."""
    # algorithmic bytes: statistics pass reads x, apply pass reads x and writes y (the single-launch path reads once)
    with _Timed("groupnorm", 0.0, 3 * 2.0 * instances * rows * c):
        ldx = int(c if ldx is None else ldx)
        _check(load().sp_groupnorm_ld_f16(_rows(x, "x", ldx).data_ptr(), ldx, _ptr(gamma), _ptr(beta),
                                          _f16(y, "y").data_ptr(), instances, rows, c, groups, eps, int(silu),
                                          ws.data_ptr(), ws.numel() * ws.element_size(), _stream()), "sp_groupnorm_f16")
    return y


def groupnorm_tile_sums(x, part, gamma, beta, y, *, instances, rows, c, groups, eps, silu, stats, ldx=None, part_b=None,
                        c_a=None):
    """GroupNorm(+SiLU) of ``x`` from the per-tile column sums its producing contraction left in ``part``
    (``gemm(..., gn_part=part)``): no statistics pass over ``x`` (``sp_groupnorm_tile_sums_f16``).  ``part_b`` / ``c_a``: x is
    the concatenation of two producers' outputs, channels ``[0, c_a)`` summed in ``part``, the rest in ``part_b``."""
    ldx = int(ldx if ldx is not None else x.stride(0))
    with _Timed("groupnorm", 0.0, 2 * 2.0 * instances * rows * c):
        _check(load().sp_groupnorm_tile_sums2_f16(_f16(x, "x").data_ptr(), ldx, part.data_ptr(), int(c_a if part_b is not None else c),
                                                  _ptr(part_b), _ptr(gamma), _ptr(beta), _f16(y, "y").data_ptr(), instances, rows,
                                                  c, groups, float(eps), int(silu), stats.data_ptr(), _stream()),
               "sp_groupnorm_tile_sums2_f16")
    return y


def groupnorm_fold_linear_tile_sums(part, gamma, beta, w, bias, w_out, bias_out, *, instances, rows, c, groups, eps, n, stats):
    """``groupnorm_fold_linear`` with the statistics folded from a producer's per-tile column sums (``gemm(..., gn_part=)``)."""
    _check(load().sp_groupnorm_fold_linear_tile_sums_f16(part.data_ptr(), _ptr(gamma), _ptr(beta), instances, rows, c, groups,
                                                         float(eps), _f16(w, "w").data_ptr(), _ptr(bias), n,
                                                         _f16(w_out, "w_out").data_ptr(), bias_out.data_ptr(),
                                                         stats.data_ptr(), _stream()),
           "sp_groupnorm_fold_linear_tile_sums_f16")


def groupnorm_fold_linear(x, gamma, beta, w, bias, w_out, bias_out, *, instances, rows, c, groups, eps, n, ws, ldx=None):
    """GroupNorm (no activation) folded into the linear layer ``w`` [n][c] behind it: ONE pass over ``x`` (statistics) and
    ``w_out`` [instances][n][c] fp16 / ``bias_out`` [instances][n] fp32 for ``gemm(x, w_out, ..., w_group_rows=rows,
    w_group_stride=n*c, bias=None, bias2=bias_out, bias2_rows=rows)`` on the RAW ``x`` (see include/svdpipe.h)."""
    with _Timed("groupnorm", 0.0, 2.0 * instances * rows * c + 2.0 * instances * n * c):
        ldx = int(c if ldx is None else ldx)
        if tuple(w_out.shape) != (instances, n, c) or not w_out.is_contiguous() or bias_out.dtype != torch.float32 \
                or tuple(bias_out.shape) != (instances, n) or not bias_out.is_contiguous():
            raise ValueError("groupnorm_fold_linear: w_out must be contiguous fp16 [instances][n][c], bias_out fp32 [instances][n]")
        _check(load().sp_groupnorm_fold_linear_f16(_rows(x, "x", ldx).data_ptr(), ldx, _ptr(gamma), _ptr(beta), instances,
                                                   rows, c, groups, eps, _f16(w, "w").data_ptr(), _ptr(bias), n,
                                                   _f16(w_out, "w_out").data_ptr(), bias_out.data_ptr(), ws.data_ptr(),
                                                   ws.numel() * ws.element_size(), _stream()), "sp_groupnorm_fold_linear_f16")
    return w_out, bias_out


def layernorm(x, gamma, beta, y, *, rows, c, eps=1e-5, addvec=None, addvec_rows=0, sum_out=None):
    with _Timed("layernorm", 0.0, (2 + (sum_out is not None)) * 2.0 * rows * c):
        _check(load().sp_layernorm_f16(_f16(x, "x").data_ptr(), _ptr(addvec), addvec_rows, _ptr(sum_out),
                                       gamma.data_ptr(), beta.data_ptr(), _f16(y, "y").data_ptr(), rows, c, eps,
                                       _stream()), "sp_layernorm_f16")
    return y


def ln_stats(x, stats, *, rows, c, eps=1e-5, addvec=None, addvec_rows=0, sum_out=None):
    """Per-row (mean, rstd) of a LayerNorm that the next GEMM applies itself (``gemm(..., ln_stats=, ln_colsum=)``)."""
    with _Timed("ln_stats", 0.0, (1 + (sum_out is not None)) * 2.0 * rows * c + 8.0 * rows):
        _check(load().sp_ln_stats_f16(_f16(x, "x").data_ptr(), _ptr(addvec), addvec_rows, _ptr(sum_out),
                                      stats.data_ptr(), rows, c, eps, _stream()), "sp_ln_stats_f16")
    return stats


def attn_spatial(q, k, v, o, *, ldq, ldk, ldv, ldo, batch, seq, heads, scale=0.125):
    with _Timed("attn_spatial", 4.0 * batch * heads * seq * seq * 64):
        _check(load().sp_attn_spatial_f16(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), ldq, ldk, ldv, ldo,
                                          batch, seq, heads, scale, zero_page(o.device).data_ptr(), _stream()),
               "sp_attn_spatial_f16")
    return o


def attn_long_ws_bytes(batch, seq, heads) -> int:
    return int(load().sp_attn_long_ws_bytes(batch, seq, heads))


def attn_spatial_long(q, k, v, o, workspace, *, ldq, ldk, ldv, ldo, batch, seq, heads, scale=0.125):
    """Spatial attention through the frozen-reference kernel for long rows (csrc/attention_long.hip; other shapes go to
    the ordinary kernel inside the call); `workspace`: tensor of >= attn_long_ws_bytes() bytes (flag words)."""
    long_rows = seq >= 4096 and seq % 256 == 0            # the call's own rule; other shapes run the ordinary kernel
    with _Timed("attn_spatial_long" if long_rows else "attn_spatial", 4.0 * batch * heads * seq * seq * 64):
        _check(load().sp_attn_spatial_long_f16(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), ldq, ldk, ldv, ldo,
                                               batch, seq, heads, scale, zero_page(o.device).data_ptr(),
                                               workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                               _stream()),
               "sp_attn_spatial_long_f16")
    return o


def attn_fp8_ws_bytes(batch, seq, heads) -> int:
    return int(load().sp_attn_fp8_ws_bytes(batch, seq, heads))


def attn_spatial_fp8(q, k, v, o, workspace, *, ldq, ldk, ldv, ldo, batch, seq, heads, scale=0.125):
    """fp8-e4m3 MFMA spatial attention (BASELINE config 5); `workspace`: uint8 tensor >= attn_fp8_ws_bytes()."""
    with _Timed("attn_spatial", 4.0 * batch * heads * seq * seq * 64):
        _check(load().sp_attn_spatial_fp8(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), ldq, ldk, ldv, ldo,
                                          batch, seq, heads, scale, workspace.data_ptr(),
                                          workspace.numel() * workspace.element_size(),
                                          zero_page(o.device).data_ptr(), _stream()),
               "sp_attn_spatial_fp8")
    return o


def attn_temporal(q, k, v, o, *, ldq, ldk, ldv, ldo, batch, frames, hw, heads, scale=0.125):
    with _Timed("attn_temporal", 4.0 * batch * hw * heads * frames * frames * 64, 4 * 2.0 * batch * frames * hw * heads * 64):
        _check(load().sp_attn_temporal_f16(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), ldq, ldk, ldv, ldo,
                                           batch, frames, hw, heads, scale, zero_page(o.device).data_ptr(), _stream()),
               "sp_attn_temporal_f16")
    return o


def pack_input(latent, image_latents, out, *, in_scale, b, frames, h, w, cpad):
    _check(load().sp_pack_input_f16(_f16(latent, "latent").data_ptr(), _f16(image_latents, "image_latents").data_ptr(),
                                    out.data_ptr(), in_scale, b, frames, h, w, cpad, _stream()), "sp_pack_input_f16")
    return out


def softmax_rows(x, *, rows, cols, ld=None):
    """In-place softmax of every row of the fp16 matrix ``x`` (the VAE mid block's attention scores)."""
    ld = int(ld if ld is not None else cols)
    with _Timed("softmax_rows", 0.0, 2 * 2.0 * rows * cols):
        _check(load().sp_softmax_rows_f16(_rows(x, "x", ld).data_ptr(), ld, rows, cols, _stream()), "sp_softmax_rows_f16")
    return x


def gemm_f32out(a, w, out32, *, m, n, k, lda=None):
    """``out32[m][n]`` (fp32, contiguous) = ``a[m][k] @ w[n][k]^T`` as raw fp32 sums (``sp_gemm_f32out_f16``): logits that
    must not pass through fp16 on their way into a softmax.  n a multiple of 256, k of 64."""
    lda = int(lda if lda is not None else a.stride(0))
    _f16(a, "a"); _f16(w, "w")
    if out32.dtype != torch.float32 or not out32.is_cuda or not out32.is_contiguous() or out32.numel() < m * n:
        raise TypeError("out32 must be a contiguous float32 HIP tensor of at least m*n elements")
    if w.dim() == 2 and w.shape[0] > 1 and w.stride(0) != k:
        raise ValueError(f"w: row stride {w.stride(0)} must equal k = {k}")
    with _Timed("gemm_f32out", 2.0 * m * n * k, 2.0 * (m * k + n * k) + 4.0 * m * n):
        _check(load().sp_gemm_f32out_f16(a.data_ptr(), lda, w.data_ptr(), out32.data_ptr(), m, n, k,
                                         zero_page(a.device).data_ptr(), _stream()), "sp_gemm_f32out_f16")
    return out32


def softmax_rows_f32(x32, out16, *, rows, cols, scale=1.0, ld=None, ldo=None):
    """``out16[r][:cols] = fp16(softmax(scale * x32[r][:cols]))``, fp32 logits in, fp32 statistics (``sp_softmax_rows_f32``).
    ``out16`` may be a float16 VIEW of ``x32`` itself (``x32.view(torch.float16)``, row pitch 2*ld halves): the
    probabilities then overwrite the front of each row's logits."""
    ld = int(ld if ld is not None else x32.stride(0))
    ldo = int(ldo if ldo is not None else out16.stride(0))
    if x32.dtype != torch.float32 or not x32.is_cuda or out16.dtype != torch.float16 or not out16.is_cuda:
        raise TypeError("x32 must be float32 and out16 float16, both on a HIP device")
    with _Timed("softmax_rows", 0.0, (4.0 + 2.0) * rows * cols):
        _check(load().sp_softmax_rows_f32(x32.data_ptr(), ld, out16.data_ptr(), ldo, rows, cols, float(scale), _stream()),
               "sp_softmax_rows_f32")
    return out16


def video_strides(t, layout):
    """(sb, sc, sf) element strides of a contiguous video tensor: "bcfhw" = (B,C,F,H,W), "nchw" = (B*F,C,H,W)."""
    if layout == "bcfhw":
        _, c, f, h, w = t.shape
        return c * f * h * w, f * h * w, h * w
    _, c, h, w = t.shape
    return None, h * w, c * h * w          # sb = F*C*hw depends on the caller's frames per batch item


def vae_pack_latent(latent, rows, *, scale, flat0, n, frames_per_item, strides, h, w, cpad):
    sb, sc, sf = strides
    _check(load().sp_vae_pack_latent_f16(_f16(latent, "latent").data_ptr(), _f16(rows, "rows").data_ptr(), scale, flat0, n,
                                         frames_per_item, sb, sc, sf, h, w, cpad, _stream()), "sp_vae_pack_latent_f16")
    return rows


def vae_frames_out(rows, weight, bias, out, *, batch, frames, h, w, flat0, frames_per_item, strides):
    sb, sc, sf = strides
    if out.dtype not in (torch.float16, torch.float32) or not out.is_contiguous():
        raise TypeError("vae_frames_out: out must be a contiguous float16 or float32 tensor")
    with _Timed("vae_frames_out", 0.0, 2.0 * batch * frames * h * w * (rows.shape[1] + 3)):
        _check(load().sp_vae_frames_out_f16(_f16(rows, "rows").data_ptr(), rows.shape[1], weight.data_ptr(), bias.data_ptr(),
                                            out.data_ptr(), int(out.dtype == torch.float32), batch, frames, h, w, flat0,
                                            frames_per_item, sb, sc, sf, _stream()), "sp_vae_frames_out_f16")
    return out


def vae_image_pack(image, rows, *, batch, h, w, cpad, flip):
    _check(load().sp_vae_image_pack_f16(_f16(image, "image").data_ptr(), _f16(rows, "rows").data_ptr(), batch, h, w, cpad,
                                        int(flip), _stream()), "sp_vae_image_pack_f16")
    return rows


def vae_latent_out(rows, out, *, batch, channels, frames, h, w, flip):
    _check(load().sp_vae_latent_out_f16(_f16(rows, "rows").data_ptr(), rows.shape[1], _f16(out, "out").data_ptr(), batch,
                                        channels, frames, h, w, int(flip), _stream()), "sp_vae_latent_out_f16")
    return out


def patchify(pixels, rows, *, batch, h, w, patch, kpad):
    _check(load().sp_patchify_f16(_f16(pixels, "pixels").data_ptr(), _f16(rows, "rows").data_ptr(), batch, h, w, patch, kpad,
                                  _stream()), "sp_patchify_f16")
    return rows


def attn_small(q, k, v, o, *, ldq, ldk, ldv, ldo, batch, seq, heads, head_dim, scale):
    with _Timed("attn_small", 4.0 * batch * heads * seq * seq * head_dim):
        _check(load().sp_attn_small_f16(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), ldq, ldk, ldv, ldo, batch, seq,
                                        heads, head_dim, scale, _stream()), "sp_attn_small_f16")
    return o


def gelu(x, y, *, quick=False):
    _check(load().sp_gelu_f16(_f16(x, "x").data_ptr(), _f16(y, "y").data_ptr(), x.numel(), int(quick), _stream()), "sp_gelu_f16")
    return y


class ClockStamps:
    """Stamps of the shader-clock counter against the constant 100 MHz counter, taken in stream order between other work
    (``sp_clock_stamp``; bench.py ``roofline.clock_ghz_live``).  ``stamp()`` enqueues one on the current stream (a ~2 us
    launch of 64 one-wave workgroups, each noting the XCD it ran on); ``ghz()`` (after a synchronise) pairs the first and the
    last stamp XCD by XCD and returns (GHz, seconds between the stamps, XCDs seen in both) -- the shader clock the chip held
    over that stretch -- or None with fewer than two stamps."""
    BLOCKS = 64

    def __init__(self, device, capacity):
        import torch
        self.buf = torch.zeros((capacity, self.BLOCKS, 4), dtype=torch.int64, device=device)
        self.n = 0

    def stamp(self):
        if self.n < self.buf.shape[0]:
            _check(load().sp_clock_stamp(self.buf[self.n].data_ptr(), self.BLOCKS, _stream()), "sp_clock_stamp")
            self.n += 1

    def ghz(self):
        if self.n < 2:
            return None
        b = self.buf[:self.n].cpu()
        real = b[:, :, 2].double().mean(dim=1)                 # when each stamp ran (its workgroups: within microseconds)
        first, last = b[int(real.argmin())], b[int(real.argmax())]
        ratios, secs = [], []
        for x in sorted(set(first[:, 0].tolist()) & set(last[:, 0].tolist())):
            f, l = first[first[:, 0] == x][0], last[last[:, 0] == x][0]
            dr = int(l[2] - f[2])
            if dr > 0:
                ratios.append(0.1 * int(l[1] - f[1]) / dr)
                secs.append(dr / 1e8)
        if not ratios:
            return None
        return sum(ratios) / len(ratios), sum(secs) / len(secs), len(ratios)


def euler_step(latent, eps_cond, eps_uncond, guidance, out, *, ld_eps, sigma, sigma_next, b, frames, h, w):
    _check(load().sp_euler_step_f16(latent.data_ptr(), eps_cond.data_ptr(), _ptr(eps_uncond), ld_eps, _ptr(guidance),
                                    out.data_ptr(), sigma, sigma_next, b, frames, h, w, _stream()), "sp_euler_step_f16")
    return out


def concat_channels(a, ca, b, cb, out, rows):
    with _Timed("concat", 0.0, 2 * 2.0 * rows * (ca + cb)):
        _check(load().sp_concat_channels_f16(a.data_ptr(), ca, b.data_ptr(), cb, out.data_ptr(), rows, _stream()),
               "sp_concat_channels_f16")
    return out


def add_rowvec(x, vec32, y, rows, c):
    _check(load().sp_add_rowvec_f16(x.data_ptr(), vec32.data_ptr(), y.data_ptr(), rows, c, _stream()), "sp_add_rowvec_f16")
    return y


def dummy_unet_forward(x, w1, b1, w2, b2, ln_w, ln_b, gain, ln_eps):
    """DummyUNet forward on a HIP device (fp32, (B,C,F,H,W)); see csrc/dummy_unet.hip."""
    B, C, F, H, W = x.shape
    hidden_c = w1.shape[0]
    out = torch.empty_like(x)
    hidden = torch.empty((B, hidden_c, F, H, W), dtype=torch.float32, device=x.device)
    _check(load().sp_dummy_unet_f32(x.data_ptr(), out.data_ptr(), hidden.data_ptr(), w1.contiguous().data_ptr(),
                                    b1.data_ptr(), w2.contiguous().data_ptr(), b2.data_ptr(), _ptr(ln_w), _ptr(ln_b),
                                    float(ln_eps), int(ln_w is not None), float(gain), B, C, hidden_c, F, H, W,
                                    _stream()), "sp_dummy_unet_f32")
    return out

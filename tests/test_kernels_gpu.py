"""GPU parity tests of every C-ABI kernel against a plain fp32 torch CPU computation of the same op
on the same fp16-rounded inputs.  Tolerances (stated per test): fp16 storage, fp32 accumulation ->
relative L2 error <= 2e-3, max abs error <= 1e-2 * max|ref| unless noted."""

import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from vdpp_amd.hip import ops
    return ops


def _w():
    from vdpp_amd.models import weights
    return weights


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def check(out, ref, l2=2e-3, mx=1e-2):
    out = out.float().cpu()
    assert torch.isfinite(out).all()
    e = rel_l2(out, ref)
    m = float((out - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    assert e <= l2 and m <= mx, f"rel_l2={e:.3e} max_rel={m:.3e}"


def rel_l2_t(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def h(t):  # fp16-rounded fp32 copy (what the kernel actually sees)
    return t.half().float()


@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (1000, 320, 320), (2016, 1280, 1280), (77, 64, 128),
                                   (4096, 960, 640), (300, 2560, 320), (5000, 256, 64), (8064, 1280, 128),
                                   (4100, 128, 640), (3000, 128, 192), (3000, 64, 64), (2600, 320, 128)])
def test_gemm_linear(m, n, k):
    ops = _ops()
    g = torch.Generator().manual_seed(m + n + k)
    a = h(torch.randn(m, k, generator=g)); w = h(torch.randn(n, k, generator=g) / math.sqrt(k))
    bias = torch.randn(n, generator=g)
    res = h(torch.randn(m, n, generator=g))
    out = torch.empty(m, n, dtype=torch.float16, device=DEV)
    ops.gemm(a.half().to(DEV), w.half().to(DEV), out, m=m, n=n, cin=k, bias=bias.to(DEV),
             res1=res.half().to(DEV), r1scale=0.5, oscale=2.0)
    ref = 2.0 * (a @ w.t() + bias) + 0.5 * res
    check(out, ref)


@pytest.fixture(params=[256, 192, 128])
def force_large_tiles(request):
    with _ops().gemm_route(2, bm=request.param):   # eligible shapes through gemm_pp.hip (BM x 256 / BM x 320 tiles)
        yield


@pytest.mark.parametrize("m,n,k", [(1024, 256, 64), (700, 512, 192), (3000, 320, 64), (2049, 960, 192), (5000, 1280, 128)])
def test_gemm_large_tile_linear(force_large_tiles, m, n, k):
    ops = _ops()
    g = torch.Generator().manual_seed(m + n + k)
    a = h(torch.randn(m, k, generator=g)); w = h(torch.randn(n, k, generator=g) / math.sqrt(k))
    bias = torch.randn(n, generator=g); b2 = torch.randn(2, n, generator=g)
    res = h(torch.randn(m, n, generator=g)); res2 = h(torch.randn(m, n, generator=g))
    out = torch.empty(m, n, dtype=torch.float16, device=DEV)
    half_rows = (m + 1) // 2
    ops.gemm(a.half().to(DEV), w.half().to(DEV), out, m=m, n=n, cin=k, bias=bias.to(DEV), bias2=b2.to(DEV),
             bias2_rows=half_rows, res1=res.half().to(DEV), r1scale=0.5, res2=res2.half().to(DEV), r2scale=-0.25, oscale=2.0)
    ref = 2.0 * (a @ w.t() + bias + b2.repeat_interleave(half_rows, 0)[:m]) + 0.5 * res - 0.25 * res2
    check(out, ref)


def test_gemm_large_tile_geglu_conv_temporal(force_large_tiles):
    ops, W = _ops(), _w()
    g = torch.Generator().manual_seed(77)
    # geglu
    m, k, inner = 1500, 128, 256
    a = h(torch.randn(m, k, generator=g)); w = h(torch.randn(2 * inner, k, generator=g) / math.sqrt(k)); b = torch.randn(2 * inner, generator=g)
    wi, bi = W.interleave_geglu(w, b)
    out = torch.empty(m, inner, dtype=torch.float16, device=DEV)
    ops.gemm(a.half().to(DEV), wi.to(DEV), out, m=m, n=2 * inner, cin=k, bias=bi.to(DEV), geglu=True)
    y = a @ w.t() + b
    check(out, y[:, :inner] * F.gelu(y[:, inner:]))
    # conv3x3 (stride 1 and the fused x2 upsample), 320 and 256 output channels
    for cin, cout, hh, ww, ups in ((64, 320, 20, 28, 0), (96, 256, 9, 11, 1)):
        x = h(torch.randn(3, cin, hh, ww, generator=g)); wc = h(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)); bc = torch.randn(cout, generator=g)
        xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
        ref = F.conv2d(xin, wc, bc, padding=1)
        ho, wo = ref.shape[2:]
        cpad = W.round_up(cin, 64)
        xp = torch.zeros(3, hh, ww, cpad); xp[..., :cin] = x.permute(0, 2, 3, 1)
        o = torch.empty(3 * ho * wo, cout, dtype=torch.float16, device=DEV)
        ops.gemm(xp.half().to(DEV), W.pack_conv3x3(wc).to(DEV), o, m=3 * ho * wo, n=W.round_up(cout, 64), cin=cpad, mode=ops.A_CONV3X3,
                 conv=(3, hh, ww, ho, wo, 1, ups), bias=bc.to(DEV), ldd=cout)
        check(o.view(3, ho, wo, cout).permute(0, 3, 1, 2), ref)
    # temporal conv
    frames, hw, c = 14, 90, 256
    x = h(torch.randn(1, c, frames, hw, 1, generator=g)); wt = h(torch.randn(c, c, 3, 1, 1, generator=g) / math.sqrt(3 * c)); bt = torch.randn(c, generator=g)
    ref = F.conv3d(x, wt, bt, padding=(1, 0, 0))
    rows = x[..., 0].permute(0, 2, 3, 1).reshape(frames * hw, c).contiguous()
    o = torch.empty_like(rows, dtype=torch.float16, device=DEV)
    ops.gemm(rows.half().to(DEV), W.pack_tconv3(wt).to(DEV), o, m=rows.shape[0], n=c, cin=c, mode=ops.A_TEMPORAL3, temporal=(frames, hw), bias=bt.to(DEV))
    check(o.view(1, frames, hw, c).permute(0, 3, 1, 2), ref[..., 0])


@pytest.mark.parametrize("bm,m,n,k,geglu", [(256, 32256, 2560, 320, True), (192, 32256, 2560, 320, True),
                                            # K >= 640, whole tiles, rows on 128-byte lines: K-step pairs as ONE whole-line image (VAR 3)
                                            (256, 32256, 2560, 640, True), (256, 32256, 1024, 640, False), (192, 16128, 2560, 1280, True),
                                            (256, 40000, 1280, 320, False), (192, 129024 // 2, 256, 640, False),
                                            (192, 50001, 960, 320, False), (192, 36000, 320, 1280, False),
                                            # 128 x 320 tiles (4 x 2 waves): every N = 320 k of the two outer levels
                                            (128, 50001, 960, 320, False), (128, 36000, 320, 1280, False),
                                            (128, 129024 // 2, 320, 320, False), (128, 33001, 640, 640, False),
                                            (128, 32256, 1920, 640, False),
                                            # 256 x 192 tiles (4 x 2 waves): the fused Q/K/V widths 960 and 1,920 (bm -192 = "bn 192")
                                            (-192, 50001, 960, 320, False), (-192, 32256, 1920, 640, False),
                                            (-192, 40000, 192, 1280, False)])
def test_gemm_persistent_stream_many_tiles(bm, m, n, k, geglu):
    """gemm_ps.hip with several tiles per workgroup (grid 256, up to 6 tiles each): the LDS-DMA stream crosses tile
    boundaries, stores of tile i are in flight behind the loads of tile i+1.  bias + bias2 (one row per batch item),
    two residuals, ragged last row tile; every output row against fp32 torch (2e-3 relative L2)."""
    ops = _ops()
    g = torch.Generator().manual_seed(m + n + k)
    a = h(torch.randn(m, k, generator=g)); w = h(torch.randn(n, k, generator=g) / math.sqrt(k))
    bias = torch.randn(n, generator=g)
    nout = n // 2 if geglu else n
    out = torch.full((m + 2, nout), 7.0, dtype=torch.float16, device=DEV)
    kw = dict(m=m, n=n, cin=k, bias=bias.to(DEV), geglu=geglu)
    y = a @ w.t() + bias
    if geglu:
        wi, bi = _w().interleave_geglu(w, bias)
        kw.update(bias=bi.to(DEV))
        y = y[:, :nout] * F.gelu(y[:, nout:])
        wdev = wi.half().to(DEV)
    else:
        b2 = torch.randn(1, n, generator=g)
        res = h(torch.randn(m, n, generator=g)); res2 = h(torch.randn(m, n, generator=g))
        kw.update(bias2=b2.to(DEV), bias2_rows=m, res1=res.half().to(DEV), r1scale=0.5, res2=res2.half().to(DEV),
                  r2scale=-0.25, oscale=2.0)
        y = 2.0 * (y + b2) + 0.5 * res - 0.25 * res2
        wdev = w.half().to(DEV)
    with (ops.gemm_route(3, bm=256, bn=192) if bm == -192 else ops.gemm_route(3, bm=bm)):
        ops.gemm(a.half().to(DEV), wdev, out[1:m + 1], **kw)
        name = ops.load().sp_gemm_last_kernel().decode()
        if bm == -192:
            assert name.startswith("gemm_ps_kernel<256, 192"), name
        if k >= 640 and name.startswith("gemm_ps_kernel<"):   # whole-line pairs exactly where launch_ps says, half-line pairs otherwise
            assert name.endswith(", 3>" if m % int(name.split("<")[1].split(",")[0]) == 0 else ", 1>"), name
        if (bm, k) == (256, 640):
            assert name.startswith("gemm_ps_kernel<256, 256") and name.endswith(", 3>"), name
    torch.cuda.synchronize()
    assert torch.all(out[0] == 7.0) and torch.all(out[m + 1] == 7.0), "guard rows written"
    check(out[1:m + 1], y)


def test_gemm_two_residuals_bias2_and_nstore():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    m, n, k = 3 * 70, 128, 192
    a = h(torch.randn(m, k, generator=g)); w = h(torch.randn(n, k, generator=g) / math.sqrt(k))
    b2 = torch.randn(3, n, generator=g)
    r1 = h(torch.randn(m, n, generator=g)); r2 = h(torch.randn(m, n, generator=g))
    out = torch.zeros(m, 100, dtype=torch.float16, device=DEV)
    # ldd = 104 (multiple of 8), only the first 100 columns are stored
    outp = torch.zeros(m, 104, dtype=torch.float16, device=DEV)
    ops.gemm(a.half().to(DEV), w.half().to(DEV), outp, m=m, n=n, cin=k, bias2=b2.to(DEV), bias2_rows=70,
             res1=r1.half().to(DEV), r1scale=0.25, res2=r2.half().to(DEV), r2scale=-1.5, oscale=0.7,
             n_store=100, ldd=104, ldr1=n, ldr2=n)
    ref = 0.7 * (a @ w.t() + b2.repeat_interleave(70, 0)) + 0.25 * r1 - 1.5 * r2
    check(outp[:, :100], ref[:, :100])
    assert float(outp[:, 100:].abs().max()) == 0.0
    del out


@pytest.mark.parametrize("m,k,inner", [(500, 128, 512), (4500, 192, 256), (3000, 64, 128)])
def test_gemm_geglu(m, k, inner):
    ops, W = _ops(), _w()
    g = torch.Generator().manual_seed(9)
    a = h(torch.randn(m, k, generator=g)); w = h(torch.randn(2 * inner, k, generator=g) / math.sqrt(k))
    b = torch.randn(2 * inner, generator=g)
    wi, bi = W.interleave_geglu(w, b)
    out = torch.empty(m, inner, dtype=torch.float16, device=DEV)
    ops.gemm(a.half().to(DEV), wi.to(DEV), out, m=m, n=2 * inner, cin=k, bias=bi.to(DEV), geglu=True)
    y = a @ w.t() + b
    ref = y[:, :inner] * F.gelu(y[:, inner:])
    check(out, ref)


@pytest.mark.parametrize("cin,cout,hh,ww,stride,ups", [(64, 64, 9, 13, 1, 0), (128, 320, 16, 24, 1, 0), (64, 128, 40, 48, 1, 0),
                                                       (64, 128, 16, 24, 2, 0), (64, 64, 7, 9, 2, 0),
                                                       (128, 64, 6, 10, 1, 1), (320, 320, 18, 32, 1, 0)])
def test_gemm_conv3x3(cin, cout, hh, ww, stride, ups):
    ops, W = _ops(), _w()
    g = torch.Generator().manual_seed(cin + cout + hh)
    nimg = 3
    x = h(torch.randn(nimg, cin, hh, ww, generator=g))
    w = h(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
    b = torch.randn(cout, generator=g)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref = F.conv2d(xin, w, b, stride=stride, padding=1)
    ho, wo = ref.shape[2:]
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    out = torch.empty(nimg * ho * wo, cout, dtype=torch.float16, device=DEV)
    ops.gemm(x_nhwc, W.pack_conv3x3(w).to(DEV), out, m=nimg * ho * wo, n=W.round_up(cout, 64), cin=cin,
             mode=ops.A_CONV3X3, conv=(nimg, hh, ww, ho, wo, stride, ups), bias=F.pad(b, (0, W.round_up(cout, 64) - cout)).to(DEV),
             n_store=cout if cout % 64 else 0, ldd=cout)
    check(out.view(nimg, ho, wo, cout).permute(0, 3, 1, 2), ref)


def test_gemm_conv_in_out_padding():
    """conv_in (Cin=8 padded to 64) and conv_out (Cout=4 padded to 64, 4 columns stored)."""
    ops, W = _ops(), _w()
    g = torch.Generator().manual_seed(3)
    x = h(torch.randn(2, 8, 10, 12, generator=g)); w = h(torch.randn(64, 8, 3, 3, generator=g) / 8)
    ref = F.conv2d(x, w, None, padding=1)
    xp = torch.zeros(2, 10, 12, 64); xp[..., :8] = x.permute(0, 2, 3, 1)
    out = torch.empty(240, 64, dtype=torch.float16, device=DEV)
    ops.gemm(xp.half().to(DEV), W.pack_conv3x3(w).to(DEV), out, m=240, n=64, cin=64, mode=ops.A_CONV3X3,
             conv=(2, 10, 12, 10, 12, 1, 0))
    check(out.view(2, 10, 12, 64).permute(0, 3, 1, 2), ref)
    w2 = h(torch.randn(4, 64, 3, 3, generator=g) / 24); b2 = torch.randn(4, generator=g)
    x2 = h(torch.randn(2, 64, 10, 12, generator=g))
    ref2 = F.conv2d(x2, w2, b2, padding=1)
    out2 = torch.empty(240, 4, dtype=torch.float16, device=DEV)
    ops.gemm(x2.permute(0, 2, 3, 1).contiguous().half().to(DEV), W.pack_conv3x3(w2).to(DEV), out2, m=240, n=64,
             cin=64, mode=ops.A_CONV3X3, conv=(2, 10, 12, 10, 12, 1, 0), bias=F.pad(b2, (0, 60)).to(DEV),
             n_store=4, ldd=4)
    check(out2.view(2, 10, 12, 4).permute(0, 3, 1, 2), ref2)


@pytest.mark.parametrize("frames,hw,c", [(14, 35, 64), (5, 128, 128), (25, 12, 64), (14, 200, 128)])
def test_gemm_temporal_conv(frames, hw, c):
    ops, W = _ops(), _w()
    g = torch.Generator().manual_seed(frames)
    bsz = 2
    x = h(torch.randn(bsz, c, frames, hw, 1, generator=g))
    w = h(torch.randn(c, c, 3, 1, 1, generator=g) / math.sqrt(3 * c)); b = torch.randn(c, generator=g)
    ref = F.conv3d(x, w, b, padding=(1, 0, 0))                        # (B,C,F,hw,1)
    rows = x[..., 0].permute(0, 2, 3, 1).reshape(bsz * frames * hw, c).contiguous()   # [(b,f,p)][c]
    out = torch.empty_like(rows, dtype=torch.float16, device=DEV)
    ops.gemm(rows.half().to(DEV), W.pack_tconv3(w).to(DEV), out, m=rows.shape[0], n=c, cin=c,
             mode=ops.A_TEMPORAL3, temporal=(frames, hw), bias=b.to(DEV))
    check(out.view(bsz, frames, hw, c).permute(0, 3, 1, 2), ref[..., 0])


@pytest.mark.parametrize("inst,rows,c,silu", [(3, 200, 64, 1), (14, 9 * 16, 320, 1), (1, 14 * 144, 640, 0),
                                              (2, 1000, 960, 1), (1, 77, 2560, 1), (5, 331, 1280, 0),
                                              # single-launch register-resident path (>= 128 (instance, group) slabs):
                                              (14, 576, 1280, 1), (14, 2304, 640, 0), (14, 576, 2560, 1),
                                              (4, 700, 1920, 1), (14, 101, 640, 1), (8, 1632, 1280, 0),
                                              (8, 1633, 1280, 0), (14, 144, 1280, 1),
                                              # small tensors normalised over all frames: single launch with 32 workgroups
                                              (1, 2016, 1280, 1), (1, 2040, 1280, 0), (1, 2041, 1280, 0), (2, 2016, 1280, 1)])
def test_groupnorm(inst, rows, c, silu):
    ops = _ops()
    g = torch.Generator().manual_seed(c + rows)
    x = h(torch.randn(inst, rows, c, generator=g) * 2 + 0.7)
    gamma = torch.randn(c, generator=g); beta = torch.randn(c, generator=g)
    ref = F.group_norm(x.permute(0, 2, 1), 32, gamma, beta, eps=1e-6)
    ref = (F.silu(ref) if silu else ref).permute(0, 2, 1)
    ws = torch.empty(ops.groupnorm_ws_bytes(inst, rows, c, 32), dtype=torch.uint8, device=DEV)
    y = torch.empty(inst, rows, c, dtype=torch.float16, device=DEV)
    ops.groupnorm(x.half().to(DEV), gamma.to(DEV), beta.to(DEV), y, instances=inst, rows=rows, c=c, groups=32,
                  eps=1e-6, silu=silu, ws=ws)
    check(y, ref)


@pytest.mark.parametrize("inst,rows,c,extra", [(2, 9216, 320, 320), (1, 14 * 2304, 640, 1280), (14, 576, 1280, 640),
                                                (14, 144, 2560, 8), (3, 1000, 960, 64)])
def test_groupnorm_strided_input(inst, rows, c, extra):
    """sp_groupnorm_ld_f16: x is the right-hand column slice of a wider buffer (the skip half of a concatenation buffer:
    rows of c + extra halves).  Every path (three-launch with and without the finalize kernel, single-launch) must give
    exactly what it gives on a dense copy, and must not touch the other columns' meaning (they hold huge values)."""
    ops = _ops()
    g = torch.Generator().manual_seed(c + rows + extra)
    x = h(torch.randn(inst * rows, c, generator=g) * 2 + 0.7)
    gamma = torch.randn(c, generator=g); beta = torch.randn(c, generator=g)
    wide = torch.full((inst * rows, c + extra), 3.0e4, dtype=torch.float16, device=DEV)
    wide[:, extra:] = x.half().to(DEV)
    ws = torch.empty(ops.groupnorm_ws_bytes(inst, rows, c, 32), dtype=torch.uint8, device=DEV)
    kw = dict(instances=inst, rows=rows, c=c, groups=32, eps=1e-6, silu=True, ws=ws)
    y_dense = torch.empty(inst * rows, c, dtype=torch.float16, device=DEV)
    ops.groupnorm(x.half().to(DEV), gamma.to(DEV), beta.to(DEV), y_dense, **kw)
    y = torch.empty_like(y_dense)
    ops.groupnorm(wide[:, extra:], gamma.to(DEV), beta.to(DEV), y, ldx=c + extra, **kw)
    assert torch.equal(y, y_dense)
    ref = F.silu(F.group_norm(x.view(inst, rows, c).permute(0, 2, 1), 32, gamma, beta, eps=1e-6)).permute(0, 2, 1)
    check(y.view(inst, rows, c), ref)
    with pytest.raises(ValueError):                       # the pitch has to be the tensor's own
        ops.groupnorm(wide[:, extra:], gamma.to(DEV), beta.to(DEV), y, ldx=c, **kw)


@pytest.mark.parametrize("inst,rows,c", [(2, 9216, 320), (1, 14 * 2304, 640), (14, 576, 1280), (14, 144, 2560),
                                          (3, 1000, 960), (1, 2016, 1280)])
def test_groupnorm_large_mean_small_variance(inst, rows, c):
    """Channels whose mean is ~50 standard deviations away from zero (real checkpoints have them): E[x^2] - mean^2 in
    fp32 would lose the variance.  Per-group offsets (some +, some -, some none) + a per-channel offset inside groups;
    three-launch path (shifted sums) and single-launch path (two passes over registers); vs fp64 group_norm on the
    fp16-rounded input.  Same tolerance as the well-conditioned cases."""
    ops = _ops()
    g = torch.Generator().manual_seed(c + rows + 1)
    cpg = c // 32
    group_off = (torch.randint(0, 3, (32,), generator=g).float() - 1.0) * 50.0          # -50, 0 or +50 sigma per group
    chan_off = group_off.repeat_interleave(cpg) + 0.3 * torch.randn(c, generator=g)
    x = h(torch.randn(inst, rows, c, generator=g) + chan_off)
    gamma = torch.randn(c, generator=g); beta = torch.randn(c, generator=g)
    ref = F.group_norm(x.double().permute(0, 2, 1), 32, gamma.double(), beta.double(), eps=1e-6)
    ref = F.silu(ref).permute(0, 2, 1).float()
    ws = torch.empty(ops.groupnorm_ws_bytes(inst, rows, c, 32), dtype=torch.uint8, device=DEV)
    y = torch.empty(inst, rows, c, dtype=torch.float16, device=DEV)
    ops.groupnorm(x.half().to(DEV), gamma.to(DEV), beta.to(DEV), y, instances=inst, rows=rows, c=c, groups=32,
                  eps=1e-6, silu=1, ws=ws)
    check(y, ref)


@pytest.mark.parametrize("inst,rows,c,n,off,pad,res", [(3, 256, 320, 320, 0.0, 0, False), (2, 512, 640, 640, 0.0, 0, True),
                                                        (4, 256, 320, 320, 50.0, 320, True), (2, 384, 1280, 256, 5.0, 0, False),
                                                        (28, 2304, 640, 640, 50.0, 0, True), (5, 1152, 64, 320, 50.0, 64, False)])
def test_groupnorm_folded_into_the_linear_layer_behind_it(inst, rows, c, n, off, pad, res):
    """proj_in(GroupNorm(x)) of a spatio-temporal transformer without the normalised tensor: sp_groupnorm_fold_linear_f16
    (one statistics pass + per-instance scaled weights, the mean term taken from the ROUNDED weights) followed by
    sp_gemm_f16 on the RAW x with w_group_rows / w_group_stride and one bias2 row per instance, against
    F.linear(F.group_norm(x)) in fp64 on the fp16-rounded inputs.  Groups 50 standard deviations off zero (the offset must
    cancel, not be rounded away), x as a column slice of a wider tensor (the skip half of a concatenation buffer), a
    residual, the output-row LayerNorm statistics (ln_out) on top, 192-row tiles (rows = 384 / 1152)."""
    ops = _ops()
    g = torch.Generator().manual_seed(inst + rows + c + n)
    cpg = c // 32
    group_off = (torch.randint(0, 3, (32,), generator=g).float() - 1.0) * off
    chan_off = group_off.repeat_interleave(cpg) + 0.3 * torch.randn(c, generator=g)
    x = h(torch.randn(inst, rows, c, generator=g) * (0.5 + torch.rand(inst, 1, 1, generator=g)) + chan_off)
    gamma, beta = 1.0 + 0.3 * torch.randn(c, generator=g), torch.randn(c, generator=g)
    w = h(torch.randn(n, c, generator=g) / math.sqrt(c))
    bias = torch.randn(n, generator=g)
    r1 = h(torch.randn(inst * rows, n, generator=g)) if res else None
    ref = F.group_norm(x.double().permute(0, 2, 1), 32, gamma.double(), beta.double(), eps=1e-6).permute(0, 2, 1)
    ref = (ref.reshape(inst * rows, c) @ w.double().t() + bias.double())
    if res:
        ref = ref + 0.5 * r1.double()
    ref = ref.float()
    xd = torch.zeros(inst * rows, c + pad, dtype=torch.float16, device=DEV)
    xd[:, :c] = x.reshape(inst * rows, c).half().to(DEV)
    xv = xd[:, :c]
    ws = torch.empty(ops.groupnorm_ws_bytes(inst, rows, c, 32), dtype=torch.uint8, device=DEV)
    w_f = torch.empty(inst, n, c, dtype=torch.float16, device=DEV)
    b_f = torch.empty(inst, n, dtype=torch.float32, device=DEV)
    ops.groupnorm_fold_linear(xv, gamma.to(DEV), beta.to(DEV), w.half().to(DEV), bias.to(DEV), w_f, b_f, instances=inst,
                              rows=rows, c=c, groups=32, eps=1e-6, n=n, ws=ws, ldx=c + pad)
    out = torch.empty(inst * rows, n, dtype=torch.float16, device=DEV)
    kw = dict(m=inst * rows, n=n, cin=c, lda=c + pad, bias2=b_f, bias2_rows=rows, w_group_rows=rows, w_group_stride=n * c)
    if res:
        kw.update(res1=r1.half().to(DEV), r1scale=0.5)
    ops.gemm(xv, w_f, out, **kw)
    # offsets of 50 sigma: |x| ~ 50 while the normalised values are ~1; the fp16 rounding of the scaled weights then shows
    # as (offset/sigma) * 2^-11 of the signal, which the rounded-weights bias removes -- same tolerance as everywhere
    check(out, ref, l2=3e-3, mx=2e-2)
    if n in (320, 640, 256):                    # and the next LayerNorm's statistics from the same call
        st = torch.empty(inst * rows, 2, dtype=torch.float32, device=DEV)
        out2 = torch.empty_like(out)
        wsl = torch.empty(inst * rows * 4, dtype=torch.float32, device=DEV) if n > 320 else None
        ops.gemm(xv, w_f, out2, ln_out=st, ln_out_eps=1e-5, workspace=wsl, **kw)
        assert torch.equal(out2, out)
        o32 = out.float()
        mean, var = o32.mean(-1), o32.var(-1, unbiased=False)
        assert float((st[:, 0] - mean).abs().max()) <= 2e-3 * float(o32.abs().max())
        assert float((st[:, 1] * torch.sqrt(var + 1e-5) - 1).abs().max()) <= 2e-3


@pytest.mark.parametrize("mode,inst,rows,cin,n,off", [(0, 3, 512, 320, 320, 0.0), (1, 4, 256, 64, 640, 3.0), (2, 2, 768, 256, 256, 0.0),
                                                      (0, 2, 1024, 128, 1280, 25.0)])
def test_groupnorm_statistics_from_the_producing_contraction(mode, inst, rows, cin, n, off):
    """sp_gemm_desc.gn_part + sp_groupnorm_tile_sums_f16: the contraction's epilogue leaves (sum, sum of squares) per 128-row
    half of every 256-row tile and output column, a small kernel folds them per (instance, group), the apply pass runs --
    against F.group_norm(+SiLU) of the contraction's stored output in fp64, for a linear layer, a 3x3 convolution (two column
    tiles) and a temporal convolution, per-instance and all-rows statistics, and a bias that puts every group 25 standard
    deviations off zero (un-shifted sums: the fold is fp64).  The column sums themselves are checked against the output."""
    ops = _ops()
    from vdpp_amd.models import weights as W
    g = torch.Generator().manual_seed(n + rows + mode)
    m = inst * rows
    taps = {0: 1, 1: 9, 2: 3}[mode]
    kw = {}
    if mode == 1:
        hh = 16; ww = rows // hh
        x = h(torch.randn(inst, hh, ww, cin, generator=g))
        wt = h(torch.randn(n, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
        wp = W.pack_conv3x3(wt.half())
        a = x.reshape(m, cin)
        kw.update(mode=ops.A_CONV3X3, conv=(inst, hh, ww, hh, ww, 1, 0))
    elif mode == 2:
        hw = rows // 3
        a = h(torch.randn(m, cin, generator=g))
        wt = h(torch.randn(n, cin, 3, 1, 1, generator=g) / math.sqrt(3 * cin))
        wp = W.pack_tconv3(wt.half())
        kw.update(mode=ops.A_TEMPORAL3, temporal=(3, hw))
    else:
        a = h(torch.randn(m, cin, generator=g))
        wp = h(torch.randn(n, cin, generator=g) / math.sqrt(cin)).half()
    bias = torch.randn(n, generator=g) + off
    bias2 = torch.randn(n, generator=g)
    out = torch.empty(m, n, dtype=torch.float16, device=DEV)
    part = torch.full((m // 256, 2, n, 2), float("nan"), dtype=torch.float32, device=DEV)
    ops.gemm(a.half().to(DEV), wp.to(DEV), out, m=m, n=n, cin=cin, bias=bias.to(DEV), bias2=bias2.to(DEV), bias2_rows=m,
             gn_part=part, **kw)
    plain = torch.empty_like(out)
    with ops.gemm_route(2, bm=256):             # the same 256-row ping-pong tiles the column sums run on
        ops.gemm(a.half().to(DEV), wp.to(DEV), plain, m=m, n=n, cin=cin, bias=bias.to(DEV), bias2=bias2.to(DEV), bias2_rows=m, **kw)
    assert torch.equal(out, plain), "asking for the column sums changed the output"
    o64 = out.double().cpu().reshape(m // 128, 128, n)
    want_s, want_q = o64.sum(1), (o64 * o64).sum(1)
    got = part.double().cpu().reshape(m // 128, n, 2)
    assert torch.isfinite(got).all()
    assert float((got[..., 0] - want_s).abs().max()) <= 2e-3 * float(want_s.abs().max() + 128.0 * 0.02)
    assert float((got[..., 1] - want_q).abs().max()) <= 3e-3 * float(want_q.abs().max())
    gamma, beta = 1.0 + 0.3 * torch.randn(n, generator=g), torch.randn(n, generator=g)
    for ni, nr in ((inst, rows), (1, m)):                       # per-instance (per-frame) and all-rows (temporal) statistics
        y = torch.empty_like(out)
        stats = torch.empty(ni * 32 * 2, dtype=torch.float32, device=DEV)
        ops.groupnorm_tile_sums(out, part, gamma.to(DEV), beta.to(DEV), y, instances=ni, rows=nr, c=n, groups=32, eps=1e-6,
                                silu=True, stats=stats)
        ref = F.silu(F.group_norm(out.double().cpu().reshape(ni, nr, n).permute(0, 2, 1), 32, gamma.double(), beta.double(),
                                  eps=1e-6)).permute(0, 2, 1).reshape(m, n).float()
        check(y, ref, l2=2e-3, mx=2e-2)
    # ---- with residuals the sums are of the FINAL stored values (the tile is rebuilt in LDS behind the stores)
    r1 = torch.randn(m, n, generator=g).half().to(DEV)
    r2 = (torch.randn(m, n, generator=g) * 2.0 + off).half().to(DEV)
    rkw = dict(res1=r1, r1scale=0.75, res2=r2, r2scale=1.0, oscale=0.5)
    out_r = torch.empty(m, n, dtype=torch.float16, device=DEV)
    part_r = torch.full((m // 256, 2, n, 2), float("nan"), dtype=torch.float32, device=DEV)
    ops.gemm(a.half().to(DEV), wp.to(DEV), out_r, m=m, n=n, cin=cin, bias=bias.to(DEV), gn_part=part_r, **rkw, **kw)
    with ops.gemm_route(2, bm=256):
        ops.gemm(a.half().to(DEV), wp.to(DEV), plain, m=m, n=n, cin=cin, bias=bias.to(DEV), **rkw, **kw)
    assert torch.equal(out_r, plain), "asking for the column sums changed the output (residual case)"
    o64 = out_r.double().cpu().reshape(m // 128, 128, n)
    got = part_r.double().cpu().reshape(m // 128, n, 2)
    assert torch.isfinite(got).all()
    assert float((got[..., 0] - o64.sum(1)).abs().max()) <= 1e-5 * float(o64.abs().sum(1).max())      # sums of the rounded values
    assert float((got[..., 1] - (o64 * o64).sum(1)).abs().max()) <= 1e-5 * float((o64 * o64).sum(1).max())
    y = torch.empty_like(out_r)
    stats = torch.empty(inst * 32 * 2, dtype=torch.float32, device=DEV)
    ops.groupnorm_tile_sums(out_r, part_r, gamma.to(DEV), beta.to(DEV), y, instances=inst, rows=rows, c=n, groups=32, eps=1e-6,
                            silu=False, stats=stats)
    ref = F.group_norm(out_r.double().cpu().reshape(inst, rows, n).permute(0, 2, 1), 32, gamma.double(), beta.double(),
                       eps=1e-6).permute(0, 2, 1).reshape(m, n).float()
    check(y, ref, l2=2e-3, mx=2e-2)
    if n in (256, 320, 640):     # and the fold into the linear layer behind the norm from the same sums
        wl = h(torch.randn(n, n, generator=g) / math.sqrt(n)).half().to(DEV)
        bl = torch.randn(n, generator=g).to(DEV)
        w_a = torch.empty(inst, n, n, dtype=torch.float16, device=DEV); b_a = torch.empty(inst, n, dtype=torch.float32, device=DEV)
        w_b = torch.empty_like(w_a); b_b = torch.empty_like(b_a)
        ws = torch.empty(ops.groupnorm_ws_bytes(inst, rows, n, 32), dtype=torch.uint8, device=DEV)
        ops.groupnorm_fold_linear(out_r, gamma.to(DEV), beta.to(DEV), wl, bl, w_a, b_a, instances=inst, rows=rows, c=n,
                                  groups=32, eps=1e-6, n=n, ws=ws, ldx=n)
        ops.groupnorm_fold_linear_tile_sums(part_r, gamma.to(DEV), beta.to(DEV), wl, bl, w_b, b_b, instances=inst, rows=rows,
                                            c=n, groups=32, eps=1e-6, n=n, stats=stats)
        assert rel_l2_t(w_b.float(), w_a.float()) <= 1e-3 and float((b_b - b_a).abs().max()) <= 2e-2 * float(b_a.abs().max())
    with pytest.raises(ops.HipKernelError, match="gn_part"):
        ops.gemm(a.half().to(DEV), wp.to(DEV), out[:, :n // 2], m=m, n=n, cin=cin, geglu=True, gn_part=part, ldd=n, **kw)
    with pytest.raises(ops.HipKernelError, match="256-row tiles"):
        ops.groupnorm_tile_sums(out, part, gamma.to(DEV), beta.to(DEV), y, instances=m // 128, rows=128, c=n, groups=32,
                                eps=1e-6, silu=True, stats=stats)


@pytest.mark.parametrize("na,nb", [(640, 320), (320, 320), (256, 512)])
def test_groupnorm_statistics_of_a_concatenation_from_two_producers(na, nb):
    """An up block's resnet normalises [hidden | skip]: two contractions write the two column ranges of one buffer, each
    leaves its column sums, and sp_groupnorm_tile_sums2_f16 folds both (per-column sums are additive, so the 30-channel
    groups of a 640 + 320 concatenation may straddle the seam) -- against F.group_norm of the buffer in fp64."""
    ops = _ops()
    g = torch.Generator().manual_seed(na + nb)
    inst, rows, cin = 2, 512, 128
    m, c = inst * rows, na + nb
    cat = torch.empty(m, c, dtype=torch.float16, device=DEV)
    parts = []
    for n, col0, with_res in ((na, 0, True), (nb, na, False)):
        a = torch.randn(m, cin, generator=g).half().to(DEV)
        w = (torch.randn(n, cin, generator=g) / math.sqrt(cin)).half().to(DEV)
        bias = (torch.randn(n, generator=g) * 2.0).to(DEV)
        part = torch.empty(m // 256, 2, n, 2, dtype=torch.float32, device=DEV)
        kw = dict(res1=torch.randn(m, n, generator=g).half().to(DEV), r1scale=1.0) if with_res else {}
        ops.gemm(a, w, cat[:, col0:col0 + n], m=m, n=n, cin=cin, bias=bias, ldd=c, gn_part=part, **kw)
        parts.append(part)
    gamma, beta = 1.0 + 0.3 * torch.randn(c, generator=g), torch.randn(c, generator=g)
    y = torch.empty(m, c, dtype=torch.float16, device=DEV)
    stats = torch.empty(inst * 32 * 2, dtype=torch.float32, device=DEV)
    ops.groupnorm_tile_sums(cat, parts[0], gamma.to(DEV), beta.to(DEV), y, instances=inst, rows=rows, c=c, groups=32, eps=1e-6,
                            silu=True, stats=stats, part_b=parts[1], c_a=na)
    ref = F.silu(F.group_norm(cat.double().cpu().reshape(inst, rows, c).permute(0, 2, 1), 32, gamma.double(), beta.double(),
                              eps=1e-6)).permute(0, 2, 1).reshape(m, c).float()
    check(y, ref, l2=2e-3, mx=2e-2)


@pytest.mark.parametrize("mode,n,cin,cin2,pad", [(1, 320, 320, 640, 0), (1, 256, 128, 64, 64), (0, 640, 256, 960, 0), (2, 320, 64, 128, 0)])
def test_gemm_extra_linear_tap(mode, n, cin, cin2, pad):
    """sp_gemm_desc.a2: behind the taps of the convolution the contraction runs on over the channels of a SECOND tensor's
    rows against the weight columns that follow -- conv2(h) + conv_shortcut(x) of a resnet as ONE call, against the two
    torch operations in fp64; a2 as a column slice of a wider buffer (lda2 > cin2), ragged-free m of several tiles, with and
    without the GroupNorm column sums of the result."""
    ops = _ops()
    from vdpp_amd.models import weights as W
    g = torch.Generator().manual_seed(n + cin + cin2)
    inst, hh, ww = 3, 16, 16
    m = inst * hh * ww
    x2 = h(torch.randn(m, cin2 + pad, generator=g))
    wsc = h(torch.randn(n, cin2, generator=g) / math.sqrt(cin2))
    bias = torch.randn(n, generator=g)
    kw = {}
    if mode == 1:
        x = h(torch.randn(inst, hh, ww, cin, generator=g))
        wc = h(torch.randn(n, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
        wp = W.pack_conv3x3(wc.half())
        a = x.reshape(m, cin)
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), wc.double(), padding=1).permute(0, 2, 3, 1).reshape(m, n)
        kw.update(mode=ops.A_CONV3X3, conv=(inst, hh, ww, hh, ww, 1, 0))
    elif mode == 2:
        a = h(torch.randn(m, cin, generator=g))
        wt = h(torch.randn(n, cin, 3, 1, 1, generator=g) / math.sqrt(3 * cin))
        wp = W.pack_tconv3(wt.half())
        x5 = a.double().reshape(1, inst, hh * ww, cin).permute(0, 3, 1, 2)                  # (1, C, frames, pixels)
        ref = F.conv2d(x5, wt.double()[:, :, :, 0, :], padding=(1, 0))[0].permute(1, 2, 0).reshape(m, n)
        kw.update(mode=ops.A_TEMPORAL3, temporal=(inst, hh * ww))
    else:
        a = h(torch.randn(m, cin, generator=g))
        wl = h(torch.randn(n, cin, generator=g) / math.sqrt(cin))
        wp = wl.half()
        ref = a.double() @ wl.double().t()
    ref = (ref + x2[:, :cin2].double() @ wsc.double().t() + bias.double()).float()
    wcat = torch.cat([wp, wsc.half()], dim=1).contiguous().to(DEV)
    x2d = x2.half().to(DEV)
    out = torch.empty(m, n, dtype=torch.float16, device=DEV)
    ops.gemm(a.half().to(DEV), wcat, out, m=m, n=n, cin=cin, bias=bias.to(DEV), a2=x2d[:, :cin2], cin2=cin2, lda2=cin2 + pad, **kw)
    check(out, ref)
    part = torch.full((m // 256, 2, n, 2), float("nan"), dtype=torch.float32, device=DEV)
    out2 = torch.empty_like(out)
    ops.gemm(a.half().to(DEV), wcat, out2, m=m, n=n, cin=cin, bias=bias.to(DEV), a2=x2d[:, :cin2], cin2=cin2, lda2=cin2 + pad,
             gn_part=part, **kw)
    assert torch.equal(out2, out)
    o64 = out.double().cpu().reshape(m // 128, 128, n)
    got = part.double().cpu().reshape(m // 128, n, 2)
    assert float((got[..., 0] - o64.sum(1)).abs().max()) <= 2e-3 * float(o64.abs().sum(1).max())
    with pytest.raises(ops.HipKernelError, match="a2"):
        ops.gemm(a.half().to(DEV), wcat, out, m=m, n=n, cin=cin, a2=x2d[:, :cin2], cin2=cin2 - 32, lda2=cin2 + pad, **kw)


def test_gemm_rejects_row_groups_that_tiles_would_straddle():
    ops = _ops()
    a = torch.zeros(640, 320, dtype=torch.float16, device=DEV)
    w = torch.zeros(2, 320, 320, dtype=torch.float16, device=DEV)
    out = torch.empty(640, 320, dtype=torch.float16, device=DEV)
    with pytest.raises(ops.HipKernelError, match="w_group_rows"):
        ops.gemm(a, w, out, m=640, n=320, cin=320, w_group_rows=320, w_group_stride=320 * 320)      # not a multiple of 128
    with pytest.raises(ops.HipKernelError, match="per-row-group"):
        ops.gemm(a, w, out[:, :160], m=640, n=320, cin=320, geglu=True, w_group_rows=256, w_group_stride=320 * 320, ldd=320)
    # a folded LayerNorm carries the column sums of ONE weight matrix: refused together with per-group weights
    st = torch.zeros(640, 2, device=DEV); cs = torch.zeros(320, device=DEV)
    with pytest.raises(ops.HipKernelError, match="folded LayerNorm"):
        ops.gemm(a, w, out, m=640, n=320, cin=320, w_group_rows=128, w_group_stride=320 * 320, ln_stats=st, ln_colsum=cs)


@pytest.mark.parametrize("m,cin,n,extras,offset", [(1000, 320, 320, "", 0.0), (129024, 320, 320, "r", 2.0), (2016, 1280, 320, "r2", 0.0),
                                                    (5000, 640, 256, "b2", 20.0), (777, 64, 320, "r", 20.0), (64512, 1280, 256, "", 0.0),
                                                    # rows of two tiles: per-tile sums through a workspace + finalize kernel
                                                    (64512, 640, 640, "r", 0.0), (3001, 2560, 640, "r2", 20.0), (1000, 320, 512, "b2", 2.0),
                                                    # ... of three and four tiles (the 576-token level's 1,280 channels)
                                                    (16128, 1280, 1280, "r", 0.0), (2500, 256, 1024, "r2", 2.0), (900, 128, 960, "", 20.0)])
def test_gemm_output_row_layernorm_statistics(m, cin, n, extras, offset):
    """sp_gemm_desc.ln_out: the epilogue leaves (mean, rstd) of every stored output row beside the output, what an
    sp_ln_stats_f16 pass over d would compute (so that the next contraction's folded LayerNorm needs no pass of its
    own).  Against that pass and against torch on the stored values; residuals, a bias2 row per image, ragged m, rows
    with a common offset of 20 standard deviations; the output itself must equal the call without ln_out, and two calls
    must agree bit for bit."""
    ops = _ops()
    g = torch.Generator().manual_seed(m + n + cin)
    a = h(torch.randn(m, cin, generator=g))
    w = h(torch.randn(n, cin, generator=g) / math.sqrt(cin))
    b = torch.randn(n, generator=g) + offset
    kw = dict(m=m, n=n, cin=cin, bias=b.to(DEV))
    if "r" in extras:
        kw.update(res1=h(torch.randn(m, n, generator=g)).half().to(DEV), r1scale=1.0)
    if "r2" in extras:
        kw.update(res2=h(torch.randn(m, n, generator=g)).half().to(DEV), r2scale=0.5)
    if "b2" in extras:
        rows_per = 1000
        kw.update(bias2=torch.randn((m + rows_per - 1) // rows_per, n, generator=g).to(DEV), bias2_rows=rows_per)
    ad, wd = a.half().to(DEV), w.half().to(DEV)
    plain = torch.empty(m, n, dtype=torch.float16, device=DEV)
    ops.gemm(ad, wd, plain, **kw)
    out = torch.empty(m, n, dtype=torch.float16, device=DEV)
    st = torch.full((m + 1, 2), 7.0, dtype=torch.float32, device=DEV)
    tiles = n // (320 if n % 320 == 0 else 256)
    if tiles > 1:
        kw["workspace"] = torch.empty(m, 2 * tiles, dtype=torch.float32, device=DEV)
    ops.gemm(ad, wd, out, ln_out=st[:m], ln_out_eps=1e-5, **kw)
    assert "gemm_pp_kernel" in ops.load().sp_gemm_last_kernel().decode()
    assert torch.all(st[m] == 7.0)
    x = out.float().cpu()
    assert rel_l2(x, plain.float().cpu()) <= 1e-3                    # (another kernel family may have produced `plain`)
    st2 = torch.empty(m, 2, dtype=torch.float32, device=DEV)
    ops.ln_stats(out, st2, rows=m, c=n, eps=1e-5)
    mean, var = x.double().mean(dim=1), x.double().var(dim=1, unbiased=False)
    want = torch.stack([mean, 1.0 / torch.sqrt(var + 1e-5)], dim=1).float()
    got = st[:m].cpu()
    assert torch.allclose(got[:, 0], want[:, 0], rtol=0, atol=2e-4 * max(1.0, abs(offset)))
    assert torch.allclose(got[:, 1], want[:, 1], rtol=2e-3 if offset else 2e-4)
    assert torch.allclose(got, st2.cpu(), rtol=2e-3 if offset else 2e-4, atol=2e-4 * max(1.0, abs(offset)))
    out_b = torch.empty_like(out); st_b = torch.empty(m, 2, dtype=torch.float32, device=DEV)
    ops.gemm(ad, wd, out_b, ln_out=st_b, ln_out_eps=1e-5, **kw)
    assert torch.equal(out_b, out) and torch.equal(st_b, st[:m])
    with pytest.raises(ops.HipKernelError, match="ln_out"):           # at most four tiles per row
        ops.gemm(ad, h(torch.randn(1920, cin, generator=g)).half().to(DEV), torch.empty(m, 1920, dtype=torch.float16, device=DEV),
                 m=m, n=1920, cin=cin, ln_out=st_b, ln_out_eps=1e-5, workspace=torch.empty(m, 12, dtype=torch.float32, device=DEV))


@pytest.mark.parametrize("m,c,n,geglu,route", [(1000, 320, 960, False, 0), (5000, 640, 1920, False, 2), (40000, 320, 2560, True, 3),
                                               (2016, 1280, 3840, False, 0), (3000, 1280, 2560, True, 2), (700, 64, 128, True, 1),
                                               (33000, 640, 1280, False, 3),
                                               # odd row counts: the persistent kernel fetches (mean, rstd) two rows per
                                               # lane and must not take them (forced route 3 and the automatic choice)
                                               (40001, 320, 1280, False, 3), (130049, 320, 1280, False, 0),
                                               (40001, 320, 2560, True, 0)])
def test_gemm_with_folded_layernorm(m, c, n, geglu, route):
    """LayerNorm folded into the next contraction (sp_ln_stats_f16 + sp_gemm_desc.ln_stats / ln_colsum): the GEMM
    runs on the UN-normalised rows with gamma-scaled weights and applies rstd*(acc - mean*colsum) + (W.beta + b)
    in its epilogue.  Every kernel family, rows with a large common offset; vs torch layer_norm -> linear (-> GEGLU)."""
    ops = _ops()
    from vdpp_amd.models.unet_hip import _Dense
    g = torch.Generator().manual_seed(m + n)
    x = h(torch.randn(m, c, generator=g) * 1.7 + 3.0 * torch.randn(m, 1, generator=g))
    gamma = 1.0 + 0.3 * torch.randn(c, generator=g); beta = 0.5 * torch.randn(c, generator=g)
    w = h(torch.randn(n, c, generator=g) / math.sqrt(c)); b = torch.randn(n, generator=g)
    layer = _Dense.fold_layernorm(w, b, gamma, beta, DEV, eps=1e-5, geglu=geglu)
    xd = x.half().to(DEV)
    st = torch.empty(m, 2, device=DEV)
    ops.ln_stats(xd, st, rows=m, c=c, eps=1e-5)
    nout = n // 2 if geglu else n
    out = torch.full((m + 2, nout), 7.0, dtype=torch.float16, device=DEV)
    with ops.gemm_route(route):
        ops.gemm(xd, layer.w, out[1:m + 1], m=m, n=n, cin=c, bias=layer.bias, geglu=geglu, ln_stats=st,
                 ln_colsum=layer.colsum)
    torch.cuda.synchronize()
    assert torch.all(out[0] == 7.0) and torch.all(out[m + 1] == 7.0)
    y = F.layer_norm(x, (c,), gamma, beta, eps=1e-5) @ w.t() + b
    if geglu:
        y = y[:, :nout] * F.gelu(y[:, nout:])
    mu, var = x.mean(1), x.var(1, unbiased=False)
    assert float((st[:, 0].cpu() - mu).abs().max()) < 2e-3 and rel_l2(st[:, 1].cpu(), (var + 1e-5).rsqrt()) < 1e-4
    check(out[1:m + 1], y, l2=3e-3, mx=2e-2)


@pytest.mark.parametrize("case", ["conv3x3", "temporal_forced", "ff2_forced", "ln_geglu_forced", "ragged_forced"])
def test_gemm_split_k(case):
    """Split-K (sp_gemm_desc.workspace): the 2,016-row level's long-K contractions on 256 x 256 tiles, K slices summed
    by the reduce kernel which also applies the epilogue.  The 3x3 convolution picks it automatically (K >= 8192); the
    forced cases (route 4) cover shorter K, the folded LayerNorm + GEGLU epilogue and a ragged m / n_store.  The result has to
    agree with fp32 torch and with the one-pass kernel on the same inputs (same rounding points: only the fp32
    summation order differs), and no byte outside the promised workspace may be written."""
    ops = _ops()
    from vdpp_amd.models import weights as Wt
    g = torch.Generator().manual_seed(len(case))
    kw, route, geglu, n_store = {}, 0, False, 0
    if case == "conv3x3":
        nimg, hh, ww, cin, n = 14, 9, 16, 1280, 1280
        x = h(torch.randn(nimg, cin, hh, ww, generator=g))
        wc = h(torch.randn(n, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
        m = nimg * hh * ww
        y = F.conv2d(x, wc, None, padding=1).permute(0, 2, 3, 1).reshape(m, n)
        a, w = x.permute(0, 2, 3, 1).contiguous().half().to(DEV), Wt.pack_conv3x3(wc).to(DEV)
        kw = dict(mode=ops.A_CONV3X3, conv=(nimg, hh, ww, hh, ww, 1, 0))
    elif case == "temporal_forced":
        route = 4
        frames, hw, cin, n = 14, 144, 1280, 1280
        m = frames * hw
        x = h(torch.randn(1, cin, frames, hw, 1, generator=g))
        w3 = h(torch.randn(n, cin, 3, 1, 1, generator=g) / math.sqrt(3 * cin))
        y = F.conv3d(x, w3, None, padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(m, n)
        a, w = x[..., 0].permute(0, 2, 3, 1).reshape(m, cin).contiguous().half().to(DEV), Wt.pack_tconv3(w3).to(DEV)
        kw = dict(mode=ops.A_TEMPORAL3, temporal=(frames, hw))
    elif case == "ff2_forced":
        route = 4
        m, cin, n = 2016, 5120, 1280
        xa = h(torch.randn(m, cin, generator=g)); wl = h(torch.randn(n, cin, generator=g) / math.sqrt(cin))
        y = xa @ wl.t()
        a, w = xa.half().to(DEV), wl.half().to(DEV)
    else:
        m, cin, n = (2016, 1280, 2560) if case == "ln_geglu_forced" else (1931, 320, 512)
        route = 4
        xa = h(torch.randn(m, cin, generator=g) + 2.0); wl = h(torch.randn(n, cin, generator=g) / math.sqrt(cin))
        a = xa.half().to(DEV)
        if case == "ln_geglu_forced":
            from vdpp_amd.models.unet_hip import _Dense
            gamma = 1.0 + 0.3 * torch.randn(cin, generator=g); beta = 0.5 * torch.randn(cin, generator=g)
            bl = torch.randn(n, generator=g)
            layer = _Dense.fold_layernorm(wl, bl, gamma, beta, DEV, eps=1e-5, geglu=True)
            st = torch.empty(m, 2, device=DEV)
            ops.ln_stats(a, st, rows=m, c=cin, eps=1e-5)
            w, geglu = layer.w, True
            kw = dict(ln_stats=st, ln_colsum=layer.colsum)
            y = F.layer_norm(xa, (cin,), gamma, beta, eps=1e-5) @ wl.t() + bl
            y = y[:, :n // 2] * F.gelu(y[:, n // 2:])
            bias = layer.bias
        else:
            w, y, n_store = wl.half().to(DEV), xa @ wl.t(), 501
            kw = dict(n_store=n_store, ldd=504)
    nout = n // 2 if geglu else n
    if case != "ln_geglu_forced":
        bt = torch.randn(n, generator=g); bias = bt.to(DEV); y = y + bt
        r1 = h(torch.randn(m, nout, generator=g)); y = y + 0.5 * r1
        kw.update(res1=r1.half().to(DEV), r1scale=0.5, ldr1=nout)
    need = ops.gemm_workspace_bytes(m=m, n=n, cin=cin, mode=kw.get("mode", ops.A_LINEAR))
    assert need > 0
    ws = torch.full((need // 4 + 256,), 55.0, device=DEV)
    ns = n_store or nout
    ldd = kw.pop("ldd", nout)
    out = torch.full((m + 2, ldd), 7.0, dtype=torch.float16, device=DEV)
    one = torch.full((m, ldd), 7.0, dtype=torch.float16, device=DEV)
    with ops.gemm_route(route):
        ops.gemm(a, w, out[1:m + 1], m=m, n=n, cin=cin, bias=bias, geglu=geglu, workspace=ws[:need // 4], ldd=ldd, **kw)
        ops.gemm(a, w, one, m=m, n=n, cin=cin, bias=bias, geglu=geglu, ldd=ldd, **kw)       # one pass (no scratch)
    torch.cuda.synchronize()
    assert torch.all(ws[need // 4:] == 55.0) and not torch.all(ws[:need // 4] == 55.0), "split-K not taken / overran"
    assert torch.all(out[0] == 7.0) and torch.all(out[m + 1] == 7.0) and torch.all(out[1:m + 1, ns:] == 7.0)
    check(out[1:m + 1, :ns], y[:, :ns], l2=3e-3, mx=2e-2)
    assert rel_l2(out[1:m + 1, :ns].float().cpu(), one[:, :ns].float().cpu()) < 6e-4


@pytest.mark.parametrize("rows,c", [(100, 64), (4099, 320), (12345, 320), (5001, 640), (4097, 960), (6002, 1280), (130, 1280)])
def test_ln_stats_rows_per_wave(rows, c):
    """sp_ln_stats_f16: several rows per wave (ragged row counts: the last wave's clamped rows must not be written),
    with and without the per-frame pre-add and the written-out sum; vs fp64 row statistics of the fp16 values."""
    ops = _ops()
    g = torch.Generator().manual_seed(rows + c)
    x = h(torch.randn(rows, c, generator=g) * 2 + 5 * torch.randn(rows, 1, generator=g))
    st = torch.full((rows + 1, 2), -7.0, device=DEV)
    ops.ln_stats(x.half().to(DEV), st[:rows], rows=rows, c=c, eps=1e-5)
    xd = x.double()
    assert torch.all(st[rows] == -7.0)
    assert float((st[:rows, 0].cpu().double() - xd.mean(1)).abs().max()) < 1e-4
    assert rel_l2(st[:rows, 1].cpu(), (xd.var(1, unbiased=False) + 1e-5).rsqrt().float()) < 1e-5
    per = 50
    add = h(torch.randn((rows + per - 1) // per, c, generator=g))
    sm = torch.full((rows + 1, c), 7.0, dtype=torch.float16, device=DEV)
    ops.ln_stats(x.half().to(DEV), st[:rows], rows=rows, c=c, eps=1e-5, addvec=add.half().to(DEV), addvec_rows=per,
                 sum_out=sm[:rows])
    xs = (x + add.repeat_interleave(per, 0)[:rows]).half()
    assert torch.equal(sm[:rows].cpu(), xs) and torch.all(sm[rows] == 7.0) and torch.all(st[rows] == -7.0)
    xsd = xs.double()
    assert float((st[:rows, 0].cpu().double() - xsd.mean(1)).abs().max()) < 1e-4
    assert rel_l2(st[:rows, 1].cpu(), (xsd.var(1, unbiased=False) + 1e-5).rsqrt().float()) < 1e-5


@pytest.mark.parametrize("rows,c", [(100, 64), (1000, 320), (513, 640), (130, 1280)])
def test_layernorm(rows, c):
    ops = _ops()
    g = torch.Generator().manual_seed(rows)
    x = h(torch.randn(rows, c, generator=g) * 3 + 1)
    gamma = torch.randn(c, generator=g); beta = torch.randn(c, generator=g)
    y = torch.empty(rows, c, dtype=torch.float16, device=DEV)
    ops.layernorm(x.half().to(DEV), gamma.to(DEV), beta.to(DEV), y, rows=rows, c=c)
    check(y, F.layer_norm(x, (c,), gamma, beta))
    # with the per-frame pre-add (frame positional embedding) and the sum written out
    per = 50
    nv = (rows + per - 1) // per
    add = h(torch.randn(nv, c, generator=g))
    s = torch.empty(rows, c, dtype=torch.float16, device=DEV)
    ops.layernorm(x.half().to(DEV), gamma.to(DEV), beta.to(DEV), y, rows=rows, c=c, addvec=add.half().to(DEV),
                  addvec_rows=per, sum_out=s)
    xs = (x + add.repeat_interleave(per, 0)[:rows]).half().float()
    check(s, xs, l2=1e-6, mx=1e-6)
    check(y, F.layer_norm(xs, (c,), gamma, beta))


@pytest.mark.parametrize("batch,seq,heads", [(2, 128, 1), (3, 200, 2), (1, 6, 1), (2, 576, 5), (1, 1000, 3), (1, 2304, 2),
                                              # >= 1,024 keys: the slot-structured 256-query kernel (ragged keys and queries)
                                              (1, 1024, 1), (2, 1100, 2), (1, 1281, 3), (1, 1025, 1), (1, 4097, 1)])
def test_attention_spatial(batch, seq, heads):
    ops = _ops()
    g = torch.Generator().manual_seed(seq)
    c = heads * 64
    qkv = h(torch.randn(batch * seq, 3 * c, generator=g))
    d = qkv.half().to(DEV)
    o = torch.empty(batch * seq, c, dtype=torch.float16, device=DEV)
    ops.attn_spatial(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], o, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=batch,
                     seq=seq, heads=heads)
    q, k, v = [t.reshape(batch, seq, heads, 64).transpose(1, 2) for t in qkv.split(c, dim=1)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(batch * seq, c)
    check(o, ref, l2=3e-3, mx=2e-2)


@pytest.mark.parametrize("seq", [700, 1500])
def test_attention_spatial_online_softmax_rescale(seq):
    """Force a late maximum: one key far down the sequence dominates (exercises the rescale branch; 1,500 keys: of the
    slot-structured kernel)."""
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    c = 64
    q = h(torch.randn(seq, c, generator=g)); k = h(torch.randn(seq, c, generator=g)); v = h(torch.randn(seq, c, generator=g))
    k[seq - 50] = q[5] * 3.0
    k[130] = q[77] * 2.0
    k[seq // 2] = q[300] * 2.5
    o = torch.empty(seq, c, dtype=torch.float16, device=DEV)
    ops.attn_spatial(q.half().to(DEV), k.half().to(DEV), v.half().to(DEV), o, ldq=c, ldk=c, ldv=c, ldo=c, batch=1, seq=seq, heads=1)
    ref = F.scaled_dot_product_attention(q[None, None], k[None, None], v[None, None])[0, 0]
    check(o, ref, l2=3e-3, mx=2e-2)


@pytest.mark.parametrize("batch,seq,heads", [(2, 4608, 2), (1, 9216, 1), (3, 4096, 1)])
def test_attention_spatial_long_rows(batch, seq, heads):
    """Rows of 4,096-9,216 tokens (the benchmark's level 0 is 9,216): 64-144 K/V tiles per query block, with late
    dominant keys so that the deferred-rescale branch runs far into the row; same tolerance as the short rows."""
    ops = _ops()
    g = torch.Generator().manual_seed(seq + heads)
    c = heads * 64
    qkv = h(torch.randn(batch * seq, 3 * c, generator=g))
    for i, (qrow, krow, amp) in enumerate([(5, seq - 50, 3.0), (77, 130, 2.0), (300, seq // 2, 2.5), (40, 3000, 3.0),
                                           (100, 1000, 2.5), (470, 2000, 2.0)]):
        qkv[krow, c:c + 64] = h(qkv[qrow, :64] * amp)          # head 0 of batch item 0
    d = qkv.half().to(DEV)
    o = torch.empty(batch * seq, c, dtype=torch.float16, device=DEV)
    ops.attn_spatial(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], o, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=batch,
                     seq=seq, heads=heads)
    q, k, v = [t.reshape(batch, seq, heads, 64).transpose(1, 2) for t in qkv.split(c, dim=1)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(batch * seq, c)
    check(o, ref, l2=3e-3, mx=2e-2)


def _attn_long(ops, d, c, batch, seq, heads, ws=None):
    o = torch.empty(batch * seq, c, dtype=torch.float16, device=DEV)
    if ws is None:
        ws = torch.full((ops.attn_long_ws_bytes(batch, seq, heads) // 4,), 0x7fffffff, dtype=torch.int32, device=DEV)  # stale junk
    ops.attn_spatial_long(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], o, ws, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c,
                          batch=batch, seq=seq, heads=heads)
    return o, ws


def _sdpa(qkv, c, batch, seq, heads):
    q, k, v = [t.reshape(batch, seq, heads, 64).transpose(1, 2) for t in qkv.split(c, dim=1)]
    return F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(batch * seq, c)


@pytest.mark.parametrize("batch,seq,heads,amp", [(2, 4608, 2, 1.0), (1, 9216, 1, 1.0), (3, 4096, 1, 1.0), (1, 4096, 2, 3.0),
                                                  (1, 5120, 1, 0.05),
                                                  # not this kernel's shapes: passed through to the ordinary kernel
                                                  (1, 4224, 1, 1.0), (2, 2304, 2, 1.0), (1, 200, 1, 1.0)])
def test_attention_spatial_long(batch, seq, heads, amp):
    """Frozen-reference kernel (csrc/attention_long.hip) vs fp32 attention: the long-row cases of the ordinary kernel with
    their late dominant keys (those blocks overflow the frozen reference and take the second pass), peaky (amp 3: scores
    of +-70) and flat (amp 0.05) rows; same tolerance.  The workspace starts with stale junk (the call zeroes it)."""
    ops = _ops()
    g = torch.Generator().manual_seed(seq + heads)
    c = heads * 64
    qkv = h(torch.randn(batch * seq, 3 * c, generator=g) * amp)
    qkv[:, 2 * c:] = h(qkv[:, 2 * c:] / amp)
    planted = []
    if seq >= 4096:
        planted = [(5, seq - 50, 3.0), (77, 130, 2.0), (300, seq // 2, 2.5), (40, 3000, 3.0), (100, 1000, 2.5)]
        for qrow, krow, a in planted:
            qkv[krow, c:c + 64] = h(qkv[qrow, :64] * a)            # head 0 of batch item 0
    o, ws = _attn_long(ops, qkv.half().to(DEV), c, batch, seq, heads)
    check(o, _sdpa(qkv, c, batch, seq, heads), l2=3e-3, mx=2e-2)
    if seq >= 4096 and seq % 256 == 0:
        flags = ws.cpu()[:batch * heads * (seq // 256)].view(batch * heads, seq // 256)     # (+ one word: flagged waves)
        assert set(flags.unique().tolist()) <= {0, 1}
        if amp == 1.0:        # unit-variance rows stay within 16 of their warm-up maximum; keys were planted in (0, 0) only
            assert int(flags[1:].sum()) == 0


@pytest.mark.parametrize("case", ["late_peaks_everywhere", "late_peak_one_block", "peak_in_own_tile", "peak_in_tile_zero",
                                  "just_below_overflow", "just_above_overflow"])
def test_attention_spatial_long_second_pass(case):
    """Rows whose largest score arrives after the warm-up and exceeds the frozen reference by more than fp16 can hold are
    flagged and recomputed by the ordinary kernel; peaks inside the warm-up (own tokens, tile 0) are not.  The result
    must not depend on which kernel produced it, and the flags must be exactly the blocks that needed it."""
    ops = _ops()
    g = torch.Generator().manual_seed(len(case))
    batch, seq, heads = 2, 4096, 2
    c = heads * 64
    nblk = seq // 256
    qkv = h(torch.randn(batch * seq, 3 * c, generator=g))
    q, k = qkv[:, :c].view(batch, seq, heads, 64), qkv[:, c:2 * c].view(batch, seq, heads, 64)
    expect = torch.zeros(batch, heads, nblk, dtype=torch.int32)
    if case == "late_peaks_everywhere":      # every query has a key far away that matches it: 0.18 * 3 * |q|^2 ~ +35
        for b in range(batch):
            for hd in range(heads):
                k[b, :, hd] = h(q[b, :, hd].roll(1500, dims=0) * 3.0)
        expect[:] = 1
    elif case == "late_peak_one_block":
        k[1, 3000, 0] = h(q[1, 700, 0] * 2.5)          # query 700 of (item 1, head 0), block 2: +29 against a warm-up
                                                       # maximum of ~5; the other queries see at most ~+13 from this key
        expect[1, 0, 700 // 256] = 1
    elif case == "peak_in_own_tile":
        k[0, 1100, 1] = h(q[0, 1030, 1] * 2.5)         # key 1100: same block as query 1030 (1024..1279): the warm-up sees it
    elif case == "peak_in_tile_zero":
        k[1, 17, 1] = h(q[1, 2222, 1] * 2.5)
    else:
        # one query, one late key, score above the row's warm-up maximum by a chosen amount (log2 units): p = 2^gap
        gap = 15.9 if case == "just_below_overflow" else 16.1
        q[0, :, 0] *= 0.05                             # flat rows: warm-up maximum ~ 0
        k[0, :, 0] *= 0.05
        unit = torch.zeros(64); unit[0] = 8.0
        q[0, 500, 0] = unit
        k[0, 3500, 0] = 0.0
        kk = k[0, :, 0].clone(); kk[3500] = 0.0
        s_all = (q[0, 500, 0] @ kk.T) * 0.125 * 1.4426950408889634
        warm = torch.cat([s_all[256:512], s_all[:64]]).max()      # own tiles of block 1 + tile 0
        k[0, 3500, 0, 0] = h((warm + gap) / (8.0 * 0.125 * 1.4426950408889634))
        expect[0, 0, 1] = 1 if gap > 16 else 0
    qkv = h(qkv)
    o, ws = _attn_long(ops, qkv.half().to(DEV), c, batch, seq, heads)
    check(o, _sdpa(qkv, c, batch, seq, heads), l2=3e-3, mx=2e-2)
    o2 = torch.empty_like(o)           # and against the ordinary kernel: same arithmetic, another reference point
    d = qkv.half().to(DEV)
    ops.attn_spatial(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], o2, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=batch, seq=seq,
                     heads=heads)
    check(o, o2.float().cpu(), l2=2e-3, mx=1e-2)
    flags = ws.cpu()[:batch * heads * nblk].view(batch, heads, nblk)
    assert torch.equal(flags, expect), f"flagged blocks {flags.nonzero().tolist()} expected {expect.nonzero().tolist()}"
    # the word behind the flag words counts the waves that flagged (the workgroups of this small call all start at once:
    # nobody bails out, every flagged block was flagged by a wave that ran to the end)
    waves = int(ws.cpu()[batch * heads * nblk])
    assert (waves > 0) == bool(expect.any()) and waves <= 4 * int(expect.sum())


def test_attention_spatial_long_worst_case_is_bounded():
    """Data on which the frozen reference fails EVERYWHERE (every query has a matching key far from its own tokens): once
    1/16 of the call's waves have flagged, workgroups that start later hand their block to the ordinary kernel at once
    instead of computing it twice.  At the UNet's level-0 size (14 x 9,216 tokens x 5 heads: 2,520 workgroups, 256 at a
    time) only the first round may have tried: the flagged-waves word stays within one round of waves, every block is
    flagged and the bytes are the ordinary kernel's.  The TIME bound that follows from it (<= 1.25x the ordinary kernel,
    was 2.1x) is a performance figure and lives in tools/bench_attn_long.py (`... :l`, profiles/r04_attention_long_worst_
    case.txt), not in the correctness suite: a wall-clock ratio on a shared or throttled GPU is not a test."""
    ops = _ops()
    batch, seq, heads = 14, 9216, 5
    c = heads * 64
    g = torch.Generator(device=DEV).manual_seed(5)
    d = torch.randn(batch * seq, 3 * c, generator=g, device=DEV, dtype=torch.float16)
    qv = d[:, :c].view(batch, seq, heads, 64)
    d[:, c:2 * c] = (qv.roll(3000, dims=1) * 3.0).reshape(batch * seq, c)          # key i+3000 = 3 x query i
    o_long = torch.empty(batch * seq, c, dtype=torch.float16, device=DEV)
    o_ord = torch.empty_like(o_long)
    ws = torch.zeros(ops.attn_long_ws_bytes(batch, seq, heads) // 4, dtype=torch.int32, device=DEV)
    kw = dict(ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=batch, seq=seq, heads=heads)

    def run_long():
        ops.attn_spatial_long(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], o_long, ws, **kw)

    def run_ord():
        ops.attn_spatial(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], o_ord, **kw)

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    t_ord, t_long = timed(run_ord), timed(run_long)
    nblk = batch * heads * (seq // 256)
    flags = ws.cpu()
    assert int(flags[:nblk].sum()) == nblk, "every block must have been handed to the ordinary kernel"
    bail = min(512, max(64, 4 * nblk // 16))
    assert int(flags[nblk]) >= bail and int(flags[nblk]) <= 4 * 256 + bail, \
        f"flagged-waves word {int(flags[nblk])}: more than the first round of workgroups tried"
    assert torch.equal(o_long, o_ord), "the second pass is the ordinary kernel: identical bytes"
    print(f"all-flagged call {t_long:.2f} ms vs ordinary kernel {t_ord:.2f} ms (x{t_long / t_ord:.2f}; informational)")


def test_attention_spatial_long_argument_errors():
    ops = _ops()
    c = 64
    d = torch.zeros(4096, 6 * c, dtype=torch.float16, device=DEV)
    small = torch.zeros(16, dtype=torch.int32, device=DEV)
    assert ops.attn_long_ws_bytes(14, 9216, 5) == (14 * 5 * 36 + 1) * 4
    with pytest.raises(ops.HipKernelError, match="workspace"):
        _attn_long(ops, d, 2 * c, 1, 4096, 2, ws=small)                   # two heads x 16 blocks need 32 + 1 words
    with pytest.raises(ops.HipKernelError, match="stride"):
        o = torch.empty(4096, c, dtype=torch.float16, device=DEV)
        ops.attn_spatial_long(d, d, d, o, small, ldq=3 * c + 4, ldk=3 * c, ldv=3 * c, ldo=c, batch=1, seq=4096, heads=1)


def _e4m3(t):
    """round-trip through OCP e4m3fn (what the fp8 path's quantiser stores), as fp32"""
    return t.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


def _attn_fp8(ops, d, c, batch, seq, heads):
    o = torch.empty(batch * seq, c, dtype=torch.float16, device=DEV)
    ws = torch.empty(ops.attn_fp8_ws_bytes(batch, seq, heads), dtype=torch.uint8, device=DEV)
    ops.attn_spatial_fp8(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], o, ws, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c,
                         batch=batch, seq=seq, heads=heads)
    return o, ws


@pytest.mark.parametrize("batch,seq,heads", [(2, 128, 1), (3, 200, 2), (1, 6, 1), (2, 576, 5), (1, 1000, 3), (1, 2304, 2)])
def test_attention_spatial_fp8(batch, seq, heads):
    """fp8-e4m3 MFMA attention (BASELINE config 5; parity path, not the default: DESIGN.md section 3).  Error budget on
    i.i.d. N(0,1) q/k/v, the worst case for a 3-bit mantissa (measured on MI355X): 2.2e-2 rel-L2 from rounding P to e4m3
    (checked against fp32 attention on the e4m3-rounded q/k/v; the ONE op-level bound, 4e-2 for any row length and
    amplitude, is what tools/fuzz_misc.py enforces; these fixed unit-variance cases stay under 2.6e-2) and 5.2e-2 in total
    once the rounding of q, k (score error ~3 %) and v is included (against fp32 attention on the unrounded inputs,
    bound 6e-2).  The <= 3e-2 of SURVEY 8c is asserted where it is meaningful, at the UNet boundary
    (test_unet_forward_fp8_attention_matches_oracle)."""
    ops = _ops()
    g = torch.Generator().manual_seed(seq + 7)
    c = heads * 64
    qkv = h(torch.randn(batch * seq, 3 * c, generator=g))
    o, _ = _attn_fp8(ops, qkv.half().to(DEV), c, batch, seq, heads)

    def sdpa(x):
        q, k, v = [t.reshape(batch, seq, heads, 64).transpose(1, 2) for t in x.split(c, dim=1)]
        return F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(batch * seq, c)
    check(o, sdpa(_e4m3(qkv)), l2=2.6e-2, mx=8e-2)
    check(o, sdpa(qkv), l2=6e-2, mx=2e-1)


def test_attention_spatial_fp8_operand_maps_exact():
    """Exact-integer check of the fp8 operand plumbing (quantiser layouts, the permuted V^T image, the
    accumulator-as-operand k order): q = 0 makes every score equal, so P is exactly uniform (2^8 in e4m3, exact)
    and the output must be the plain mean of V over the keys; V holds small integers that e4m3 represents
    exactly and that differ per (key, channel), so any key/channel mix-up changes the result."""
    ops = _ops()
    batch, seq, heads = 2, 192, 2
    c = heads * 64
    key = torch.arange(seq).view(1, seq, 1, 1)
    ch = torch.arange(64).view(1, 1, 1, 64)
    hd = torch.arange(heads).view(1, 1, heads, 1)
    bb = torch.arange(batch).view(batch, 1, 1, 1)
    v = (((key * 5 + ch * 3 + hd * 7 + bb * 11) % 15) - 7).float()          # integers in [-7, 7]
    # weight one key per 64-tile more: k.q must stay 0 (q = 0), so instead vary V only; also a second case below
    qkv = torch.zeros(batch * seq, 3 * c)
    qkv[:, 2 * c:] = v.reshape(batch * seq, c)
    qkv[:, c:2 * c] = torch.randn(batch * seq, c, generator=torch.Generator().manual_seed(3))
    o, _ = _attn_fp8(ops, qkv.half().to(DEV), c, batch, seq, heads)
    ref = v.mean(dim=1, keepdim=True).expand(batch, seq, heads, 64).reshape(batch * seq, c)
    check(o, ref.half().float(), l2=1e-3, mx=2e-3)
    # one-hot attention: query i matches key perm[i] only (scores 0 vs 64*8*8/8 = 512 apart) -> output = V[perm[i]]
    seq2 = 128
    perm = torch.randperm(seq2, generator=torch.Generator().manual_seed(5))
    code = torch.zeros(seq2, 64)
    for i in range(seq2):                      # 7-bit code of the key index in +-8 (exact in e4m3)
        for bit in range(7):
            code[i, bit * 8:(bit + 1) * 8] = 8.0 if (i >> bit) & 1 else -8.0
    v2 = ((torch.arange(seq2).view(seq2, 1) * 3 + torch.arange(64).view(1, 64) * 5) % 13 - 6).float()
    qkv2 = torch.cat([code[perm], code, v2], dim=1)
    o2, _ = _attn_fp8(ops, qkv2.half().to(DEV), 64, 1, seq2, 1)
    check(o2, v2[perm], l2=1e-3, mx=2e-3)


def test_attention_spatial_fp8_long_tail():
    """A peaked row with a long tail of small probabilities: unscaled, every p < 2^-9 would flush to zero in e4m3
    and the tail's share of the output (here about half) would vanish; the kernel converts p * 2^8."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    seq, c = 4096, 64
    q = h(torch.randn(seq, c, generator=g)) * 0.25
    k = h(torch.randn(seq, c, generator=g)) * 0.25
    v = h(torch.randn(seq, c, generator=g))
    k[17] = q[3] * 16.6                         # one dominant key for query 3 (p ~ 0.5); 4095 tail keys at ~1e-4 each
    v[:, 0] = 4.0                               # channel 0: +4 on the tail, -4 on the dominant key
    v[17, 0] = -4.0
    qkv = torch.cat([q, k, v], dim=1)
    o, _ = _attn_fp8(ops, qkv.half().to(DEV), c, 1, seq, 1)
    x = _e4m3(qkv)
    ref = F.scaled_dot_product_attention(x[None, None, :, :c], x[None, None, :, c:2 * c], x[None, None, :, 2 * c:])[0, 0]
    check(o, ref, l2=2e-2, mx=8e-2)
    p3 = torch.softmax((x[3, :c] @ x[:, c:2 * c].T) / 8.0, dim=0)
    assert 0.2 < float(p3[17]) < 0.8 and float(p3.topk(2).values[1]) < 2 ** -9   # the case is what it claims to be
    assert abs(float(ref[3, 0])) < 2.5                    # tail and peak both matter (a flushed tail would give -4)
    assert abs(float(o[3, 0].float().cpu()) - float(ref[3, 0])) <= 0.2


def test_attention_fp8_argument_errors():
    ops = _ops()
    from vdpp_amd import hip
    d = torch.zeros(64, 192, dtype=torch.float16, device=DEV)
    o = torch.empty(64, 64, dtype=torch.float16, device=DEV)
    small = torch.empty(16, dtype=torch.uint8, device=DEV)
    with pytest.raises(RuntimeError, match="workspace too small"):
        ops.attn_spatial_fp8(d[:, :64], d[:, 64:128], d[:, 128:], o, small, ldq=192, ldk=192, ldv=192, ldo=64,
                             batch=1, seq=64, heads=1)
    assert ops.attn_fp8_ws_bytes(1, 64, 1) == 3 * 64 * 64 and ops.attn_fp8_ws_bytes(2, 65, 3) == 2 * 2 * 65 * 192 + 2 * 3 * 64 * 128
    assert ops.attn_fp8_ws_bytes(0, 64, 1) == 0


@pytest.mark.parametrize("batch,frames,hw,heads", [(1, 14, 37, 1), (2, 14, 64, 5), (1, 25, 50, 2), (1, 3, 24, 1), (1, 16, 9, 2)])
def test_attention_temporal(batch, frames, hw, heads):
    ops = _ops()
    g = torch.Generator().manual_seed(frames * hw)
    c = heads * 64
    rows = batch * frames * hw
    qkv = h(torch.randn(rows, 3 * c, generator=g))
    d = qkv.half().to(DEV)
    o = torch.empty(rows, c, dtype=torch.float16, device=DEV)
    ops.attn_temporal(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], o, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=batch,
                      frames=frames, hw=hw, heads=heads)
    q, k, v = [t.reshape(batch, frames, hw, heads, 64).permute(0, 2, 3, 1, 4) for t in qkv.split(c, dim=1)]
    ref = F.scaled_dot_product_attention(q, k, v)              # (b, hw, heads, F, 64)
    ref = ref.permute(0, 3, 1, 2, 4).reshape(rows, c)
    check(o, ref, l2=3e-3, mx=2e-2)


def test_gemv_batched():
    """sp_gemv_batched_f16: a batch of same-shape GEMVs in one launch, shared input (x_stride 0) and per-problem input."""
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    batch, rows, n, k = 5, 3, 200, 320
    w = h(torch.randn(batch, n, k, generator=g) / math.sqrt(k)); b = torch.randn(batch, n, generator=g)
    x_shared = h(torch.randn(rows, k, generator=g)); x_each = h(torch.randn(batch, rows, k, generator=g))
    y32 = torch.empty(batch, rows, n, dtype=torch.float32, device=DEV)
    ops.gemv_batched(x_shared.half().to(DEV), w.half().to(DEV), b.to(DEV), batch=batch, n=n, k=k, rows=rows, x_stride=0, y32=y32)
    check(y32, torch.einsum("rk,gnk->grn", x_shared, w) + b[:, None, :], l2=1e-5, mx=1e-5)
    y16 = torch.empty(batch, rows, n, dtype=torch.float16, device=DEV)
    ops.gemv_batched(x_each.half().to(DEV), w.half().to(DEV), None, batch=batch, n=n, k=k, rows=rows, y16=y16, silu_out=True)
    check(y16, F.silu(torch.einsum("grk,gnk->grn", x_each, w)))
    # n not a multiple of the 8 outputs a wave owns: nothing may be written past the last output
    n2 = 203
    w2 = h(torch.randn(n2, k, generator=g) / math.sqrt(k))
    flat = torch.full((rows * n2 + 8,), 7.0, dtype=torch.float32, device=DEV)
    ops.gemv(x_shared.half().to(DEV), w2.half().to(DEV), None, n=n2, k=k, rows=rows, y32=flat[:rows * n2].view(rows, n2))
    check(flat[:rows * n2].view(rows, n2), x_shared @ w2.t(), l2=1e-5, mx=1e-5)
    assert torch.all(flat[rows * n2:] == 7.0)


def test_pack_input_and_euler():
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    b, f, hh, ww = 2, 5, 6, 7
    lat = h(torch.randn(b, 4, f, hh, ww, generator=g) * 30); img = h(torch.randn(b, 4, f, hh, ww, generator=g))
    out = torch.empty(b * f * hh * ww, 64, dtype=torch.float16, device=DEV)
    sigma, sigma_next = 31.5, 20.25
    scale = 1.0 / math.sqrt(sigma * sigma + 1)
    ops.pack_input(lat.half().to(DEV), img.half().to(DEV), out, in_scale=scale, b=b, frames=f, h=hh, w=ww, cpad=64)
    ref = torch.cat([(lat * scale).half().float(), img], dim=1).permute(0, 2, 3, 4, 1).reshape(-1, 8)
    check(out[:, :8], ref, l2=1e-3, mx=2e-3)
    assert float(out[:, 8:].abs().max()) == 0.0
    eps_c = h(torch.randn(b * f * hh * ww, 4, generator=g)); eps_u = h(torch.randn(b * f * hh * ww, 4, generator=g))
    new = torch.empty_like(lat, dtype=torch.float16, device=DEV)

    def euler(e_rows):
        e = e_rows.reshape(b, f, hh, ww, 4).permute(0, 4, 1, 2, 3)
        x0 = e * (-sigma / math.sqrt(sigma ** 2 + 1)) + lat / (sigma ** 2 + 1)
        return lat + (lat - x0) / sigma * (sigma_next - sigma)

    ops.euler_step(lat.half().to(DEV), eps_c.half().to(DEV), None, None, new, ld_eps=4, sigma=sigma, sigma_next=sigma_next,
                   b=b, frames=f, h=hh, w=ww)
    check(new, euler(eps_c), l2=1e-3, mx=2e-3)
    gs = torch.linspace(1.0, 3.0, f)
    ops.euler_step(lat.half().to(DEV), eps_c.half().to(DEV), eps_u.half().to(DEV), gs.to(DEV), new, ld_eps=4, sigma=sigma,
                   sigma_next=sigma_next, b=b, frames=f, h=hh, w=ww)
    gsr = gs.half().float().repeat_interleave(hh * ww).repeat(b)[:, None]
    check(new, euler(eps_u + gsr * (eps_c - eps_u)), l2=2e-3, mx=4e-3)


def test_concat_addvec_gemv_sinusoid():
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    a = h(torch.randn(100, 64, generator=g)); b = h(torch.randn(100, 128, generator=g))
    out = torch.empty(100, 192, dtype=torch.float16, device=DEV)
    ops.concat_channels(a.half().to(DEV), 64, b.half().to(DEV), 128, out, 100)
    assert torch.equal(out.cpu().float(), torch.cat([a, b], 1))
    vec = torch.randn(64, generator=g)
    y = torch.empty(100, 64, dtype=torch.float16, device=DEV)
    ops.add_rowvec(a.half().to(DEV), vec.to(DEV), y, 100, 64)
    check(y, a + vec, l2=1e-3, mx=2e-3)
    x = h(torch.randn(2, 320, generator=g)); w = h(torch.randn(1000, 320, generator=g) / 18); bb = torch.randn(1000, generator=g)
    y32 = torch.empty(2, 1000, dtype=torch.float32, device=DEV)
    ops.gemv(x.half().to(DEV), w.half().to(DEV), bb.to(DEV), n=1000, k=320, rows=2, y32=y32, silu_in=True)
    check(y32, F.silu(x).half().float() @ w.t() + bb, l2=1e-3, mx=2e-3)
    vals = torch.tensor([1.63777, 5.0, 127.0, 0.02, 0.0, 13.0])
    so = torch.empty(6, 320, dtype=torch.float16, device=DEV)
    ops.sinusoid(vals.to(DEV), so, 6, 320)
    half = 160
    fr = torch.exp(-math.log(10000.0) * torch.arange(half) / half)
    ref = torch.cat([torch.cos(vals[:, None] * fr), torch.sin(vals[:, None] * fr)], -1)
    check(so, ref, l2=2e-3, mx=2e-3)


def test_dummy_unet_hip_matches_reference_golden(golden_dir):
    """HIP DummyUNet vs the vectors minted from the reference (fp32, tolerance 1e-4 relative)."""
    from vdpp_amd.models import DummyUNet
    for name, (c, hid) in (("dummy_c8h16.npz", (8, 16)), ("dummy_c4h64.npz", (4, 64))):
        z = np.load(f"{golden_dir}/{name}")
        model = DummyUNet(c, hid)
        model.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")})
        model = model.to(DEV)
        lat = torch.from_numpy(z["input"]).to(DEV)
        with torch.no_grad():
            for s in z["timesteps"].tolist():
                lat = model(lat, s)
        check(lat, torch.from_numpy(z["final"]), l2=1e-5, mx=1e-4)


def test_clock_stamps_bracket_work():
    """sp_clock_stamp (bench.py `roofline.clock_ghz_live`): two stamps around ~tens of ms of contractions give a shader clock
    between idle and the 2.4 GHz the peak assumes, over a window that matches the host's clock, on at least one XCD."""
    import time
    ops = _ops()
    a = torch.randn(8192, 1280, device=DEV, dtype=torch.float16)
    w = torch.randn(5120, 1280, device=DEV, dtype=torch.float16) * 0.02
    out = torch.empty(8192, 5120, device=DEV, dtype=torch.float16)
    for _ in range(3):
        ops.gemm(a, w, out, m=8192, n=5120, cin=1280)
    st = ops.ClockStamps(torch.device(DEV), 4)
    assert st.ghz() is None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st.stamp()
    for _ in range(200):
        ops.gemm(a, w, out, m=8192, n=5120, cin=1280)
    st.stamp()
    torch.cuda.synchronize()
    host = time.perf_counter() - t0
    ghz, secs, xcds = st.ghz()
    assert 0.3 < ghz <= 2.6, ghz
    assert 1 <= xcds <= 8
    assert 0.5 * host < secs < 1.05 * host + 1e-3, (secs, host)
    assert ops.load().sp_clock_stamp(0, 64, 0) != 0          # a null pointer is refused, nothing is launched
    assert ops.load().sp_clock_stamp(st.buf.data_ptr(), 0, 0) != 0

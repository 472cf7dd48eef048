"""Per-kernel-family checks at the REAL shapes of BASELINE config 2 (SVD, 14 frames, latent 72x128): the level-0 and
level-3 implicit GEMMs, the 9,216-token attention, the full-size normalisations.  The CPU oracle cannot run these
sizes in seconds, so each output is compared on a random SAMPLE OF ROWS with an fp32 torch evaluation of the same
rows on the GPU (gather + matmul in fp32; no CPU run).  Tolerance 2e-3 relative L2 per op (3e-3 attention), as for the
reduced-size oracle tests."""

import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from vdpp_amd.hip import ops
    return ops


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _rows(m, k, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randperm(m, generator=g)[:k].sort().values.to(DEV)


@pytest.mark.parametrize("frames,h,w,cin,cout", [(14, 72, 128, 320, 320),      # level 0: 129,024 x 320 x 2,880
                                                 (14, 9, 16, 2560, 1280),      # level 3: 2,016 x 1,280 x 23,040
                                                 (14, 36, 64, 640, 640)])
def test_conv3x3_full_shape_row_sample(frames, h, w, cin, cout):
    ops = _ops()
    from vdpp_amd.models import weights as W
    torch.manual_seed(cin + h)
    x = torch.randn(frames, h, w, cin, device=DEV).half()
    wc = (torch.randn(cout, cin, 3, 3, device=DEV) / math.sqrt(9 * cin)).half()
    bias = torch.randn(cout, device=DEV)
    m = frames * h * w
    res = torch.randn(m, cout, device=DEV).half()
    out = torch.empty(m, cout, dtype=torch.float16, device=DEV)
    ops.gemm(x.reshape(m, cin), W.pack_conv3x3(wc), out, m=m, n=cout, cin=cin, mode=ops.A_CONV3X3,
             conv=(frames, h, w, h, w, 1, 0), bias=bias, res1=res, r1scale=1.0)
    rows = _rows(m, 96, 1)
    f, rem = rows // (h * w), rows % (h * w)
    yy, xx = rem // w, rem % w
    xp = F.pad(x.float(), (0, 0, 1, 1, 1, 1))                         # zero halo, NHWC
    patches = torch.stack([xp[f, yy + ky, xx + kx] for ky in range(3) for kx in range(3)], dim=1)   # [rows][9][cin]
    wk = wc.float().permute(0, 2, 3, 1).reshape(cout, 9, cin)        # tap-major, like the packed weight
    ref = torch.einsum("rtc,ntc->rn", patches, wk) + bias + res[rows].float()
    assert rel_l2(out[rows].float(), ref) <= 2e-3


def test_temporal_conv_full_shape_row_sample():
    ops = _ops()
    from vdpp_amd.models import weights as W
    frames, hw, c = 14, 72 * 128, 320
    torch.manual_seed(5)
    m = frames * hw
    x = torch.randn(m, c, device=DEV).half()
    wt = (torch.randn(c, c, 3, 1, 1, device=DEV) / math.sqrt(3 * c)).half()
    res = torch.randn(m, c, device=DEV).half()
    out = torch.empty(m, c, dtype=torch.float16, device=DEV)
    ops.gemm(x, W.pack_tconv3(wt), out, m=m, n=c, cin=c, mode=ops.A_TEMPORAL3, temporal=(frames, hw),
             bias=None, res1=res, r1scale=1.0, oscale=0.5)
    rows = _rows(m, 128, 2)
    f = rows // hw
    acc = torch.zeros(len(rows), c, device=DEV)
    for tap in range(3):
        src = rows + (tap - 1) * hw
        ok = ((f + tap - 1) >= 0) & ((f + tap - 1) < frames)
        a = torch.where(ok[:, None], x[src.clamp(0, m - 1)].float(), torch.zeros((), device=DEV))
        acc += a @ wt[:, :, tap, 0, 0].float().t()
    ref = 0.5 * acc + res[rows].float()
    assert rel_l2(out[rows].float(), ref) <= 2e-3


@pytest.mark.parametrize("m,n,k,geglu,nres", [(129024, 2560, 320, True, 0),      # FF1 at level 0 (persistent-stream kernel)
                                              (129024, 320, 1280, False, 2),     # FF2 at level 0 with both residuals
                                              (129024, 960, 320, False, 0),      # QKV at level 0
                                              (2016, 10240, 1280, True, 0),      # FF1 at level 3 (small-tile kernels)
                                              (32256, 5120, 640, True, 0)])
def test_linear_full_shape_row_sample(m, n, k, geglu, nres):
    ops = _ops()
    from vdpp_amd.models import weights as W
    torch.manual_seed(n + k)
    a = torch.randn(m, k, device=DEV).half()
    w = (torch.randn(n, k, device=DEV) / math.sqrt(k)).half()
    bias = torch.randn(n, device=DEV)
    nout = n // 2 if geglu else n
    out = torch.empty(m, nout, dtype=torch.float16, device=DEV)
    kw = {}
    if nres >= 1:
        kw.update(res1=torch.randn(m, nout, device=DEV).half(), r1scale=0.75)
    if nres >= 2:
        kw.update(res2=torch.randn(m, nout, device=DEV).half(), r2scale=0.25, oscale=0.75)
    if geglu:
        wi, bi = W.interleave_geglu(w, bias)
        ops.gemm(a, wi, out, m=m, n=n, cin=k, bias=bi, geglu=True)
    else:
        ops.gemm(a, w, out, m=m, n=n, cin=k, bias=bias, **kw)
    rows = _rows(m, 128, 3)
    y = a[rows].float() @ w.float().t() + bias
    if geglu:
        y = y[:, :nout] * F.gelu(y[:, nout:])
    y = y * kw.get("oscale", 1.0)
    if nres >= 1:
        y = y + 0.75 * kw["res1"][rows].float()
    if nres >= 2:
        y = y + 0.25 * kw["res2"][rows].float()
    assert rel_l2(out[rows].float(), y) <= 2e-3


@pytest.mark.parametrize("seq,heads", [(9216, 5), (2304, 10)])
def test_attention_full_shape_query_sample(seq, heads):
    """14 frames x `seq` tokens x `heads` heads of 64 (the level-0 / level-1 spatial attention): a sample of
    (frame, head, query) rows against softmax(q K^T / 8) V in fp32."""
    ops = _ops()
    frames, c = 14, heads * 64
    torch.manual_seed(seq)
    qkv = torch.randn(frames * seq, 3 * c, device=DEV).half()
    o = torch.empty(frames * seq, c, dtype=torch.float16, device=DEV)
    ops.attn_spatial(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], o, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c,
                     batch=frames, seq=seq, heads=heads)
    g = torch.Generator().manual_seed(9)
    worst = 0.0
    for _ in range(12):
        f = int(torch.randint(0, frames, (1,), generator=g)); hd = int(torch.randint(0, heads, (1,), generator=g))
        qs = torch.randint(0, seq, (16,), generator=g).to(DEV)
        blk = qkv[f * seq:(f + 1) * seq].float()
        q = blk[qs, hd * 64:(hd + 1) * 64]
        k = blk[:, c + hd * 64:c + (hd + 1) * 64]
        v = blk[:, 2 * c + hd * 64:2 * c + (hd + 1) * 64]
        ref = torch.softmax(q @ k.t() * 0.125, dim=-1) @ v
        worst = max(worst, rel_l2(o[f * seq + qs, hd * 64:(hd + 1) * 64].float(), ref))
    assert worst <= 3e-3


@pytest.mark.parametrize("inst,rows,c,temporal", [(14, 9216, 320, False), (1, 14 * 9216, 320, True), (14, 2304, 1280, False)])
def test_groupnorm_full_shape(inst, rows, c, temporal):
    ops = _ops()
    torch.manual_seed(c)
    x = (torch.randn(inst, rows, c, device=DEV) * 1.5 + 0.4).half()
    gamma = torch.randn(c, device=DEV); beta = torch.randn(c, device=DEV)
    ws = torch.empty(ops.groupnorm_ws_bytes(inst, rows, c, 32), dtype=torch.uint8, device=DEV)
    y = torch.empty_like(x)
    ops.groupnorm(x, gamma, beta, y, instances=inst, rows=rows, c=c, groups=32, eps=1e-6, silu=1, ws=ws)
    ref = F.silu(F.group_norm(x.float().permute(0, 2, 1), 32, gamma, beta, eps=1e-6)).permute(0, 2, 1)
    assert rel_l2(y.float(), ref) <= 2e-3


def test_layernorm_full_shape():
    ops = _ops()
    rows, c = 129024, 320
    torch.manual_seed(3)
    x = torch.randn(rows, c, device=DEV).half()
    gamma = torch.randn(c, device=DEV); beta = torch.randn(c, device=DEV)
    y = torch.empty_like(x)
    ops.layernorm(x, gamma, beta, y, rows=rows, c=c, eps=1e-5)
    ref = F.layer_norm(x.float(), (c,), gamma, beta, eps=1e-5)
    assert rel_l2(y.float(), ref) <= 2e-3


@pytest.mark.parametrize("inst,rows,c", [(28, 9216, 320), (28, 2304, 640)])
def test_groupnorm_fold_linear_full_shape_two_videos(inst, rows, c):
    """The transformer-entry GroupNorm folded into proj_in at the shapes bench.py's two-videos-per-call configuration
    runs: 28 instances (2 videos x 14 frames) x 9,216 / 2,304 rows, one scaled weight copy per instance picked by the
    contraction's tiles (w_group_rows).  Row sample against fp32 F.linear(F.group_norm(x)) evaluated on the GPU; every
    instance has its own scale and offset so that a tile reading a neighbouring instance's weights fails."""
    ops = _ops()
    torch.manual_seed(inst + c)
    scale = 0.5 + 1.5 * torch.rand(inst, 1, 1, device=DEV)
    off = torch.randn(inst, 1, c, device=DEV)
    x = (torch.randn(inst, rows, c, device=DEV) * scale + off).half()
    gamma = 1.0 + 0.3 * torch.randn(c, device=DEV); beta = torch.randn(c, device=DEV)
    w = (torch.randn(c, c, device=DEV) / math.sqrt(c)).half()
    bias = torch.randn(c, device=DEV)
    m = inst * rows
    ws = torch.empty(ops.groupnorm_ws_bytes(inst, rows, c, 32), dtype=torch.uint8, device=DEV)
    w_f = torch.empty(inst, c, c, dtype=torch.float16, device=DEV)
    b_f = torch.empty(inst, c, dtype=torch.float32, device=DEV)
    ops.groupnorm_fold_linear(x.reshape(m, c), gamma, beta, w, bias, w_f, b_f, instances=inst, rows=rows, c=c, groups=32,
                              eps=1e-6, n=c, ws=ws, ldx=c)
    out = torch.empty(m, c, dtype=torch.float16, device=DEV)
    res = torch.randn(m, c, device=DEV).half()
    ops.gemm(x.reshape(m, c), w_f, out, m=m, n=c, cin=c, bias2=b_f, bias2_rows=rows, w_group_rows=rows,
             w_group_stride=c * c, res1=res, r1scale=1.0)
    normed = F.group_norm(x.float().permute(0, 2, 1), 32, gamma, beta, eps=1e-6).permute(0, 2, 1).reshape(m, c)
    sel = _rows(m, 256, 7)
    # every instance is in the sample's reach: add the first and last row of each
    edge = torch.cat([torch.arange(inst, device=DEV) * rows, torch.arange(inst, device=DEV) * rows + rows - 1])
    sel = torch.cat([sel, edge]).unique()
    ref = normed[sel] @ w.float().t() + bias + res[sel].float()
    assert rel_l2(out[sel].float(), ref) <= 3e-3

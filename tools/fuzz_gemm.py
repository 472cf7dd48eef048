#!/usr/bin/env python3
"""Randomised cross-check of sp_gemm_f16 against fp32 torch (CPU): shapes, modes and epilogue flags drawn at random,
output rows guarded on both sides (any write outside [0, m) x [0, n_store) fails).
usage: fuzz_gemm.py [cases] [seed] [route] [bm]   (route / bm: sp_gemm_set_route, as in tests/test_fuzz_gpu.py;
route 4 = split-K: few-row shapes with N % 256 == 0 drawn more often and a workspace handed to every call)"""
import math, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import vdpp_amd  # noqa
from vdpp_amd.hip import ops
from vdpp_amd.models import weights as W

DEV = "cuda"
GUARD = 3


def h(t):
    return t.half().float()


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def one(rng, g, stream_shapes=False, splitk=False):
    mode = rng.choice([0, 0, 1, 2])
    n = rng.choice([64, 128, 192, 256, 320, 512, 640, 960, 1280])
    cin = rng.choice([64, 128, 192, 320])
    if splitk and rng.random() < 0.8:
        n, cin = rng.choice([256, 512, 1280]), rng.choice([128, 320, 640])
    if stream_shapes and rng.random() < 0.7:          # what gemm_ps.hip takes: linear, K >= 256, N % 256 or % 320
        mode, n, cin = 0, rng.choice([192, 256, 320, 512, 640, 960, 1280, 1920, 2560]), rng.choice([320, 640, 1280])
    geglu = mode == 0 and n % 128 == 0 and rng.random() < 0.3
    kw = {}
    if mode == 0:
        m = rng.choice([1, 7, 64, 255, 256, 257, 1000, 2560, 2561, 2700, 3000, 4097, 6001])
        if splitk:
            m = rng.choice([1, 7, 255, 256, 257, 1000, 2016, 2560])
        if stream_shapes and rng.random() < 0.5:
            m = rng.choice([12000, 20001, 33000, 48000])     # several tiles per workgroup of a 256-workgroup grid
        a = h(torch.randn(m, cin, generator=g))
        wt = h(torch.randn(n, cin, generator=g) / math.sqrt(cin))
        y = a @ wt.t()
        a_dev, w_dev = a.half().to(DEV), wt.half().to(DEV)
    elif mode == 1:
        nimg, hh, ww = rng.choice([1, 2, 5]), rng.choice([3, 8, 17, 24]), rng.choice([4, 9, 16, 40])
        if splitk:
            nimg, hh, ww = rng.choice([1, 2, 5]), rng.choice([3, 8, 9]), rng.choice([4, 9, 16])
        stride, ups = rng.choice([(1, 0), (1, 0), (2, 0), (1, 1)])
        x = h(torch.randn(nimg, cin, hh, ww, generator=g))
        wc = h(torch.randn(n, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
        xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
        ref = F.conv2d(xin, wc, None, padding=1, stride=stride)
        ho, wo = ref.shape[2:]
        m = nimg * ho * wo
        y = ref.permute(0, 2, 3, 1).reshape(m, n)
        a_dev = x.permute(0, 2, 3, 1).contiguous().half().to(DEV)
        w_dev = W.pack_conv3x3(wc).to(DEV)
        kw = dict(mode=ops.A_CONV3X3, conv=(nimg, hh, ww, ho, wo, stride, ups))
    else:
        frames, hw = rng.choice([3, 14, 25]), rng.choice([5, 64, 150])
        m = frames * hw
        x = h(torch.randn(1, cin, frames, hw, 1, generator=g))
        wt3 = h(torch.randn(n, cin, 3, 1, 1, generator=g) / math.sqrt(3 * cin))
        y = F.conv3d(x, wt3, None, padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(m, n)
        a_dev = x[..., 0].permute(0, 2, 3, 1).reshape(m, cin).contiguous().half().to(DEV)
        w_dev = W.pack_tconv3(wt3).to(DEV)
        kw = dict(mode=ops.A_TEMPORAL3, temporal=(frames, hw))
    bias = torch.randn(n, generator=g) if rng.random() < 0.7 else None
    if geglu:
        # interleave rows so that (h, gate) pairs sit 16 apart, as the engine packs GEGLU weights
        wi, bi = W.interleave_geglu(wt, bias if bias is not None else torch.zeros(n))
        w_dev = wi.to(DEV)
        yb = y + (bias if bias is not None else 0)
        inner = n // 2
        y = yb[:, :inner] * F.gelu(yb[:, inner:])
        bias_dev = bi.to(DEV) if bias is not None else None
        nout = inner
    else:
        if bias is not None:
            y = y + bias
        bias_dev = bias.to(DEV) if bias is not None else None
        nout = n
    if rng.random() < 0.3 and not geglu:
        rows_per = rng.choice([1, 3, 64, max(1, m // 2)])
        nb = (m + rows_per - 1) // rows_per
        b2 = torch.randn(nb, n, generator=g)
        y = y + b2.repeat_interleave(rows_per, 0)[:m]
        kw.update(bias2=b2.to(DEV), bias2_rows=rows_per)
    oscale = rng.choice([1.0, 1.0, 0.5, 2.0])
    y = y * oscale
    if rng.random() < 0.5:
        r1 = h(torch.randn(m, nout, generator=g)); s1 = rng.choice([1.0, 0.5, -0.25])
        y = y + s1 * r1
        kw.update(res1=r1.half().to(DEV), r1scale=s1)
        if rng.random() < 0.3:
            r2 = h(torch.randn(m, nout, generator=g)); s2 = rng.choice([1.0, 0.75])
            y = y + s2 * r2
            kw.update(res2=r2.half().to(DEV), r2scale=s2)
    n_store = nout
    if rng.random() < 0.15 and not geglu:
        n_store = rng.randrange(1, nout + 1)
    ldd = ((n_store + 7) // 8) * 8 + rng.choice([0, 8])
    if n_store != nout:
        kw.update(n_store=n_store)
        if "res1" in kw: kw["ldr1"] = nout
        if "res2" in kw: kw["ldr2"] = nout
    if splitk:                         # scratch with a guard behind it (slabs must stay inside the promised bytes)
        need = ops.gemm_workspace_bytes(m=m, n=n, cin=cin, mode=mode)
        wsbuf = torch.full((need // 4 + 64,), 123.0, dtype=torch.float32, device=DEV)
        kw.update(workspace=wsbuf[:need // 4] if need else None)
    buf = torch.full((m + 2 * GUARD, ldd), 7.0, dtype=torch.float16, device=DEV)
    out = buf[GUARD:GUARD + m]
    ops.gemm(a_dev, w_dev, out, m=m, n=n, cin=cin, bias=bias_dev, geglu=geglu, oscale=oscale, ldd=ldd, **kw)
    torch.cuda.synchronize()
    got = buf.float().cpu()
    if splitk:
        assert torch.all(wsbuf[need // 4:] == 123.0), "split-K slabs overran the workspace"
    desc = f"mode={mode} m={m} n={n} cin={cin} geglu={geglu} n_store={n_store} ldd={ldd} flags={sorted(k for k in kw if k not in ('mode', 'conv', 'temporal'))}"
    assert torch.all(got[:GUARD] == 7.0) and torch.all(got[GUARD + m:] == 7.0), "guard rows written: " + desc
    assert torch.all(got[GUARD:GUARD + m, n_store:] == 7.0), "columns past n_store written: " + desc
    e = rel_l2(got[GUARD:GUARD + m, :n_store], y[:, :n_store])
    assert torch.isfinite(got).all() and e <= 3e-3, f"rel_l2={e:.3e}: " + desc
    return e


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    route = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    bm = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    worst = 0.0
    with ops.gemm_route(route, bm=bm):
        for i in range(cases):
            worst = max(worst, one(rng, g, stream_shapes=route == 3, splitk=route == 4))
            if (i + 1) % 25 == 0:
                print(f"{i + 1} cases ok, worst rel_l2 {worst:.2e}", flush=True)
    print(f"fuzz_gemm: {cases} cases passed (seed {seed}, route {route}, bm {bm}), worst rel_l2 {worst:.2e}")


if __name__ == "__main__":
    main()

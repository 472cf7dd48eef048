#!/usr/bin/env python3
"""Randomised cross-checks of the attention (UNet, CLIP), normalisation, LayerNorm-statistics, GEMV, row-softmax and GELU
kernels against fp32 / fp64 torch (CPU), with guard rows around every output (no write outside the rows the call owns).
usage: fuzz_misc.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import vdpp_amd  # noqa
from vdpp_amd.hip import ops

DEV = "cuda"
G = 2


def h(t):
    return t.half().float()


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def guarded(rows, cols):
    buf = torch.full((rows + 2 * G, cols), 9.0, dtype=torch.float16, device=DEV)
    return buf, buf[G:G + rows]


def check_guard(buf, rows, what):
    got = buf.float().cpu()
    assert torch.all(got[:G] == 9.0) and torch.all(got[G + rows:] == 9.0), "guard rows written: " + what
    assert torch.isfinite(got).all(), "non-finite output: " + what
    return got[G:G + rows]


def attn_spatial(rng, g, fp8):
    batch, heads = rng.choice([1, 2, 5]), rng.choice([1, 2, 5])
    seq = rng.choice([1, 5, 63, 64, 65, 127, 128, 129, 200, 577, 1000, 1153])
    c = heads * 64
    amp = rng.choice([0.5, 1.0, 2.0])
    qkv = h(torch.randn(batch * seq, 3 * c, generator=g) * amp)
    d = qkv.half().to(DEV)
    buf, o = guarded(batch * seq, c)
    kw = dict(ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=batch, seq=seq, heads=heads)
    if fp8:
        ws = torch.empty(ops.attn_fp8_ws_bytes(batch, seq, heads), dtype=torch.uint8, device=DEV)
        ops.attn_spatial_fp8(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], o, ws, **kw)
        x = qkv.clamp(-448, 448).to(torch.float8_e4m3fn).float()
        # ONE op-level bound for every row length and amplitude (DESIGN.md section 3, fp8 paragraph): 4e-2 relative L2
        # against fp32 attention of the e4m3-rounded inputs.  (2.2e-2 is what rounding P to e4m3 costs on long rows of
        # unit-variance inputs; peaked softmaxes and short rows average fewer rounded probabilities per output.)  The
        # 3e-2 of SURVEY 8c is asserted at the UNet boundary, where it is meaningful.
        tol = 4e-2
    else:
        ops.attn_spatial(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], o, **kw)
        x, tol = qkv, 3e-3
    torch.cuda.synchronize()
    q, k, v = [t.reshape(batch, seq, heads, 64).transpose(1, 2) for t in x.split(c, dim=1)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(batch * seq, c)
    what = f"attn_spatial{'_fp8' if fp8 else ''} batch={batch} seq={seq} heads={heads}"
    e = rel_l2(check_guard(buf, batch * seq, what), ref)
    assert e <= tol, f"rel_l2={e:.3e}: {what}"
    return e


def attn_long(rng, g):
    """Frozen-reference kernel: row lengths on both sides of its contract, amplitudes from flat to peaky, and (half of
    the cases) late keys that match a query far above its warm-up maximum, so that blocks take the second pass."""
    batch, heads = rng.choice([1, 2]), rng.choice([1, 2, 5])
    seq = rng.choice([4096, 4352, 4608, 5120, 4224, 4097, 2304])
    c = heads * 64
    amp = rng.choice([0.25, 1.0, 1.0, 2.0, 3.0])
    qkv = h(torch.randn(batch * seq, 3 * c, generator=g) * amp)
    qkv[:, 2 * c:] = h(qkv[:, 2 * c:] / amp)
    planted = rng.random() < 0.5
    if planted:
        x = qkv.view(batch, seq, 3 * heads, 64)
        for _ in range(rng.choice([1, 3, 20])):
            b, hd, qi, ki = rng.randrange(batch), rng.randrange(heads), rng.randrange(seq), rng.randrange(seq)
            x[b, ki, heads + hd] = h(x[b, qi, hd] * rng.choice([2.0, 3.0]) / max(amp * amp, 0.25))
    d = qkv.half().to(DEV)
    buf, o = guarded(batch * seq, c)
    ws = torch.full((ops.attn_long_ws_bytes(batch, seq, heads) // 4 + 1,), -1, dtype=torch.int32, device=DEV)
    ops.attn_spatial_long(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], o, ws, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=batch,
                          seq=seq, heads=heads)
    torch.cuda.synchronize()
    q, k, v = [t.reshape(batch, seq, heads, 64).transpose(1, 2) for t in qkv.split(c, dim=1)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(batch * seq, c)
    what = f"attn_spatial_long batch={batch} seq={seq} heads={heads} amp={amp} planted={planted}"
    e = rel_l2(check_guard(buf, batch * seq, what), ref)
    assert e <= 3e-3, f"rel_l2={e:.3e}: {what}"
    assert int(ws[-1]) == -1, "flag words written past the workspace: " + what
    return e


def attn_temporal(rng, g):
    batch, heads, frames, hw = rng.choice([1, 2]), rng.choice([1, 3, 5]), rng.choice([1, 2, 14, 16, 17, 25, 32]), rng.choice([1, 3, 50, 257])
    c = heads * 64
    rows = batch * frames * hw
    qkv = h(torch.randn(rows, 3 * c, generator=g))
    d = qkv.half().to(DEV)
    buf, o = guarded(rows, c)
    ops.attn_temporal(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], o, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=batch,
                      frames=frames, hw=hw, heads=heads)
    torch.cuda.synchronize()
    q, k, v = [t.reshape(batch, frames, hw, heads, 64).permute(0, 2, 3, 1, 4) for t in qkv.split(c, dim=1)]
    ref = F.scaled_dot_product_attention(q, k, v).permute(0, 3, 1, 2, 4).reshape(rows, c)
    what = f"attn_temporal batch={batch} frames={frames} hw={hw} heads={heads}"
    e = rel_l2(check_guard(buf, rows, what), ref)
    assert e <= 3e-3, f"rel_l2={e:.3e}: {what}"
    return e


def groupnorm(rng, g):
    inst = rng.choice([1, 2, 5, 14])
    rows = rng.choice([1, 7, 100, 144, 576, 577, 1000, 1632, 1633, 2304])
    c = rng.choice([64, 320, 640, 960, 1280, 1920, 2560])
    silu = rng.random() < 0.5
    eps = rng.choice([1e-5, 1e-6])
    x = h(torch.randn(inst, rows, c, generator=g) * rng.choice([0.3, 2.0]) + rng.choice([0.0, 0.7, -3.0]))
    gamma = torch.randn(c, generator=g); beta = torch.randn(c, generator=g)
    ref = F.group_norm(x.permute(0, 2, 1), 32, gamma, beta, eps=eps)
    ref = (F.silu(ref) if silu else ref).permute(0, 2, 1).reshape(inst * rows, c)
    ws = torch.empty(ops.groupnorm_ws_bytes(inst, rows, c, 32), dtype=torch.uint8, device=DEV)
    buf, y = guarded(inst * rows, c)
    ops.groupnorm(x.half().to(DEV).reshape(inst * rows, c), gamma.to(DEV), beta.to(DEV), y, instances=inst, rows=rows, c=c,
                  groups=32, eps=eps, silu=silu, ws=ws)
    torch.cuda.synchronize()
    what = f"groupnorm inst={inst} rows={rows} c={c} silu={silu}"
    e = rel_l2(check_guard(buf, inst * rows, what), ref)
    assert e <= 3e-3, f"rel_l2={e:.3e}: {what}"
    return e


def layernorm(rng, g):
    rows = rng.choice([1, 3, 64, 257, 1000, 4099])
    c = rng.choice([64, 320, 640, 1280])
    x = h(torch.randn(rows, c, generator=g) * 2 + 0.5)
    gamma = torch.randn(c, generator=g); beta = torch.randn(c, generator=g)
    buf, y = guarded(rows, c)
    ops.layernorm(x.half().to(DEV), gamma.to(DEV), beta.to(DEV), y, rows=rows, c=c)
    torch.cuda.synchronize()
    what = f"layernorm rows={rows} c={c}"
    e = rel_l2(check_guard(buf, rows, what), F.layer_norm(x, (c,), gamma, beta))
    assert e <= 3e-3, f"rel_l2={e:.3e}: {what}"
    return e


def ln_stats(rng, g):
    rows = rng.choice([1, 3, 257, 4095, 4097, 8191, 8193, 9000, 20001])      # around the row-count thresholds of the
    c = rng.choice([64, 320, 640, 960, 1280])                                 # lane-group kernels (8192 / 4096 / 2048)
    x = h(torch.randn(rows, c, generator=g) * rng.choice([0.5, 3.0]) + 4.0 * torch.randn(rows, 1, generator=g))
    add = rng.random() < 0.4
    per = rng.choice([1, 50, 333])
    xd = x.half().to(DEV)
    st = torch.full((rows + 2, 2), -7.0, device=DEV)
    kw = {}
    if add:
        av = h(torch.randn((rows + per - 1) // per, c, generator=g))
        buf, sm = guarded(rows, c)
        kw = dict(addvec=av.half().to(DEV), addvec_rows=per, sum_out=sm)
        x = (x + av.repeat_interleave(per, 0)[:rows]).half().float()
    ops.ln_stats(xd, st[1:rows + 1], rows=rows, c=c, eps=1e-5, **kw)
    torch.cuda.synchronize()
    what = f"ln_stats rows={rows} c={c} add={add}"
    assert torch.all(st[0] == -7.0) and torch.all(st[rows + 1] == -7.0), "guard rows written: " + what
    if add:
        assert torch.equal(check_guard(buf, rows, what), x), "sum_out: " + what
    xd64 = x.double()
    got = st[1:rows + 1].cpu().double()
    assert float((got[:, 0] - xd64.mean(1)).abs().max()) < 1e-4, what
    e = rel_l2(got[:, 1], (xd64.var(1, unbiased=False) + 1e-5).rsqrt())
    assert e <= 1e-5, f"rstd rel_l2={e:.3e}: {what}"
    return e


def gemv(rng, g):
    batch, rows = rng.choice([1, 1, 3, 12]), rng.choice([1, 1, 2, 14])
    n, k = rng.choice([1, 7, 8, 9, 200, 203, 320, 1280, 5120]), rng.choice([64, 320, 1024, 1280])
    w = h(torch.randn(batch, n, k, generator=g) / k ** 0.5); b = torch.randn(batch, n, generator=g)
    shared = rng.random() < 0.5
    x = h(torch.randn(rows, k, generator=g)) if shared else h(torch.randn(batch, rows, k, generator=g))
    silu_in, silu_out = rng.random() < 0.3, rng.random() < 0.3
    xin = F.silu(x).half().float() if silu_in else x
    ref = (torch.einsum("rk,gnk->grn", xin, w) if shared else torch.einsum("grk,gnk->grn", xin, w)) + b[:, None, :]
    if silu_out: ref = F.silu(ref)
    flat = torch.full((batch * rows * n + 16,), 7.0, dtype=torch.float32, device=DEV)
    y = flat[8:8 + batch * rows * n].view(batch, rows, n)
    ops.gemv_batched(x.half().to(DEV), w.half().to(DEV), b.to(DEV), batch=batch, n=n, k=k, rows=rows,
                     x_stride=0 if shared else None, y32=y, silu_in=silu_in, silu_out=silu_out)
    torch.cuda.synchronize()
    what = f"gemv batch={batch} rows={rows} n={n} k={k} shared={shared} silu={silu_in}/{silu_out}"
    assert torch.all(flat[:8] == 7.0) and torch.all(flat[8 + batch * rows * n:] == 7.0), "guard written: " + what
    e = rel_l2(y.cpu(), ref)
    assert e <= 2e-3, f"rel_l2={e:.3e}: {what}"
    return e


def attn_small(rng, g):
    """CLIP-style attention: short sequences, any head width that is a multiple of 8 (K, V of a head must fit in LDS)."""
    import math
    batch, heads = rng.choice([1, 2, 3]), rng.choice([1, 2, 5, 16])
    hd = rng.choice([8, 40, 64, 80, 128])
    seq = rng.choice([1, 2, 63, 64, 65, 200, 257, 300])
    if seq * (hd + 1) * 4 > 140 * 1024:
        seq = 130
    c = heads * hd
    qkv = h(torch.randn(batch * seq, 3 * c, generator=g) * rng.choice([0.5, 1.0, 2.0]))
    d = qkv.half().to(DEV)
    buf, o = guarded(batch * seq, c)
    ops.attn_small(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], o, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=batch, seq=seq,
                   heads=heads, head_dim=hd, scale=1.0 / math.sqrt(hd))
    torch.cuda.synchronize()
    q, k, v = [t.reshape(batch, seq, heads, hd).transpose(1, 2) for t in qkv.split(c, dim=1)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(batch * seq, c)
    what = f"attn_small batch={batch} seq={seq} heads={heads} hd={hd}"
    e = rel_l2(check_guard(buf, batch * seq, what), ref)
    assert e <= 2e-3, f"rel_l2={e:.3e}: {what}"
    return e


def softmax_rows(rng, g):
    """The VAE mid block's in-place row softmax: any column count that is a multiple of 8 up to 16,384, row pitch >= it."""
    rows = rng.choice([1, 3, 64, 257])
    cols = rng.choice([8, 64, 264, 2048, 2056, 4104, 9216, 16384])
    ld = cols + rng.choice([0, 8, 64])
    x = h(torch.randn(rows, ld, generator=g) * rng.choice([0.5, 4.0, 30.0]))
    buf = torch.full((rows + 2 * G, ld), 9.0, dtype=torch.float16, device=DEV)
    buf[G:G + rows] = x.half().to(DEV)
    ops.softmax_rows(buf[G:G + rows], rows=rows, cols=cols, ld=ld)
    torch.cuda.synchronize()
    what = f"softmax_rows rows={rows} cols={cols} ld={ld}"
    got = check_guard(buf, rows, what)
    assert torch.equal(got[:, cols:], x[:, cols:]), "columns past cols touched: " + what
    ref = torch.softmax(x[:, :cols], dim=-1)
    assert float((got[:, :cols] - ref).abs().max()) <= 2e-3, what
    e = rel_l2(got[:, :cols], ref)
    assert e <= 2e-3, f"rel_l2={e:.3e}: {what}"
    return e


def gelu(rng, g):
    n = 8 * rng.choice([1, 37, 4096, 40001])
    x = h(torch.randn(n, generator=g) * rng.choice([0.5, 3.0, 8.0]))
    quick = rng.random() < 0.5
    buf = torch.full((n + 16,), 9.0, dtype=torch.float16, device=DEV)
    ops.gelu(x.half().to(DEV), buf[8:8 + n], quick=quick)
    torch.cuda.synchronize()
    got = buf.float().cpu()
    assert torch.all(got[:8] == 9.0) and torch.all(got[8 + n:] == 9.0), "gelu guard"
    ref = x * torch.sigmoid(1.702 * x) if quick else F.gelu(x)
    assert float((got[8:8 + n] - ref).abs().max()) <= 8e-3 and rel_l2(got[8:8 + n], ref) <= 1e-3, f"gelu n={n} quick={quick}"
    return 0.0


def one(rng, g):
    kind = rng.choice(["as", "as", "as8", "at", "gn", "gn", "ln", "lns", "lns", "gv", "asm", "asm", "sm", "gl", "al"])
    if kind == "al": return attn_long(rng, g)
    if kind == "asm": return attn_small(rng, g)
    if kind == "sm": return softmax_rows(rng, g)
    if kind == "gl": return gelu(rng, g)
    if kind == "lns": return ln_stats(rng, g)
    if kind == "gv": return gemv(rng, g)
    if kind == "as": return attn_spatial(rng, g, False)
    if kind == "as8": return attn_spatial(rng, g, True)
    if kind == "at": return attn_temporal(rng, g)
    if kind == "gn": return groupnorm(rng, g)
    return layernorm(rng, g)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng, g = random.Random(seed), torch.Generator().manual_seed(seed)
    for i in range(cases):
        one(rng, g)
        if (i + 1) % 25 == 0:
            print(f"{i + 1} cases ok", flush=True)
    print(f"fuzz_misc: {cases} cases passed (seed {seed})")


if __name__ == "__main__":
    main()

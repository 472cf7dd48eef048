"""CLIP image encoder on the first stage: HIP engine vs ``transformers.CLIPVisionModelWithProjection``
itself -- the reference's own dependency (/root/reference/scripts/generate_video_demo.py:108-112, :251-254), present in
this image, run on the host in fp32 with the same (fp16-rounded, random) weights.  Through the C ABI."""

import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _ops():
    from vdpp_amd.hip import ops
    return ops


def test_patchify_matches_unfold():
    ops = _ops()
    g = torch.Generator().manual_seed(0)
    b, hh, ww, p, kpad = 2, 56, 42, 14, 640
    px = torch.randn(b, 3, hh, ww, generator=g).half()
    rows = torch.full((b * (hh // p) * (ww // p), kpad), 5.0, dtype=torch.float16, device=DEV)
    ops.patchify(px.to(DEV), rows, batch=b, h=hh, w=ww, patch=p, kpad=kpad)
    torch.cuda.synchronize()
    want = F.unfold(px.float(), kernel_size=p, stride=p).transpose(1, 2).reshape(-1, 3 * p * p)   # k = c*P*P + ky*P + kx
    assert torch.equal(rows[:, :3 * p * p].float().cpu(), want) and float(rows[:, 3 * p * p:].abs().max()) == 0.0
    with pytest.raises(ops.HipKernelError):
        ops.patchify(px.to(DEV), rows, batch=b, h=hh, w=ww, patch=p, kpad=512)


@pytest.mark.parametrize("batch,seq,heads,hd", [(2, 257, 16, 80), (1, 50, 3, 64), (1, 280, 2, 128), (1, 512, 2, 40), (3, 1, 2, 8), (1, 65, 1, 40)])
def test_attention_small(batch, seq, heads, hd):
    ops = _ops()
    g = torch.Generator().manual_seed(seq + hd)
    c = heads * hd
    qkv = (torch.randn(batch * seq, 3 * c, generator=g) * 1.5).half()
    d = qkv.to(DEV)
    o = torch.full((batch * seq + 2, c), 7.0, dtype=torch.float16, device=DEV)
    ops.attn_small(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], o[1:], ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=batch, seq=seq,
                   heads=heads, head_dim=hd, scale=1.0 / math.sqrt(hd))
    torch.cuda.synchronize()
    q, k, v = [t.float().reshape(batch, seq, heads, hd).transpose(1, 2) for t in qkv.split(c, dim=1)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(batch * seq, c)
    got = o.float().cpu()
    assert torch.all(got[0] == 7.0) and torch.all(got[-1] == 7.0)
    assert rel_l2(got[1:-1], ref) <= 2e-3
    with pytest.raises(ops.HipKernelError):          # more than 512 keys
        ops.attn_small(d, d, d, o, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=1, seq=513, heads=1, head_dim=8, scale=1.0)
    with pytest.raises(ops.HipKernelError):          # K and V of a head beyond the 160 KB of LDS
        ops.attn_small(d, d, d, o, ldq=1024, ldk=1024, ldv=1024, ldo=1024, batch=1, seq=512, heads=1, head_dim=128, scale=1.0)


@pytest.mark.parametrize("quick", [False, True])
def test_gelu(quick):
    ops = _ops()
    x = (torch.randn(257 * 128, generator=torch.Generator().manual_seed(2)) * 3).half()
    y = torch.empty_like(x, device=DEV)
    ops.gelu(x.to(DEV), y, quick=quick)
    ref = x.float() * torch.sigmoid(1.702 * x.float()) if quick else F.gelu(x.float())
    assert float((y.float().cpu() - ref).abs().max()) <= 4e-3 and rel_l2(y.float().cpu(), ref) <= 1e-3


def _pair(cfg):
    from transformers import CLIPVisionModelWithProjection
    from vdpp_amd.models.clip_hip import CLIPVisionHIP, CLIPVisionSpec

    torch.manual_seed(1234)
    ref = CLIPVisionModelWithProjection(cfg).eval()
    # transformers initialises LayerNorm to (1, 0) and biases to 0: perturb them so that every parameter matters
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for name, p in ref.named_parameters():
            if name.endswith("bias") or "norm" in name or "layrnorm" in name:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
        for p in ref.parameters():
            p.copy_(p.half().float())                 # both sides see the same fp16-representable weights
    sd = {k: v.half() for k, v in ref.state_dict().items()}
    return CLIPVisionHIP(CLIPVisionSpec.from_config(cfg), sd, DEV), ref


def test_clip_matches_committed_golden_vector(golden_dir):
    """HIP engine vs tests/golden/clip_tiny.npz (weights, pixel_values and image_embeds minted by transformers itself,
    tests/golden/make_clip_golden.py): no third-party code involved at test time."""
    import os
    import numpy as np
    from tests.golden.make_clip_golden import CFG
    from vdpp_amd.models.clip_hip import CLIPVisionHIP, CLIPVisionSpec

    z = np.load(os.path.join(golden_dir, "clip_tiny.npz"))
    spec = CLIPVisionSpec(CFG["hidden_size"], CFG["intermediate_size"], CFG["num_hidden_layers"], CFG["num_attention_heads"],
                          CFG["image_size"], CFG["patch_size"], CFG["projection_dim"], CFG["layer_norm_eps"], CFG["hidden_act"])
    sd = {k[2:]: torch.from_numpy(z[k]).half() for k in z.files if k.startswith("w:")}
    hip = CLIPVisionHIP(spec, sd, DEV)
    got = hip(torch.from_numpy(z["pixel_values"]).half().to(DEV))
    torch.cuda.synchronize()
    assert rel_l2(got.float().cpu(), torch.from_numpy(z["image_embeds"])) <= 1e-2


def test_clip_small_config_matches_transformers():
    from transformers import CLIPVisionConfig
    cfg = CLIPVisionConfig(hidden_size=128, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2, image_size=56,
                           patch_size=14, projection_dim=64, hidden_act="quick_gelu")
    hip, ref = _pair(cfg)
    px = torch.randn(2, 3, 56, 56, generator=torch.Generator().manual_seed(7)).half()
    got = hip(px.to(DEV))
    torch.cuda.synchronize()
    with torch.no_grad():
        want = ref(px.float()).image_embeds
    assert got.shape == (2, 64) and got.dtype == torch.float16 and torch.isfinite(got).all()
    assert rel_l2(got.float().cpu(), want) <= 1e-2


def test_clip_vit_h14_matches_transformers():
    """The SVD image encoder's own architecture (OpenCLIP ViT-H/14: 632 M parameters, 257 tokens x 1280, 16 heads of 80,
    32 layers, gelu, projection to 1024) with random weights; image_embeds rel-L2 <= 1e-2 against transformers in fp32."""
    from transformers import CLIPVisionConfig
    from vdpp_amd.models.clip_hip import CLIPVisionSpec, encode_image_embeddings
    s = CLIPVisionSpec.svd()
    cfg = CLIPVisionConfig(hidden_size=s.hidden_size, intermediate_size=s.intermediate_size, num_hidden_layers=s.num_hidden_layers,
                           num_attention_heads=s.num_attention_heads, image_size=s.image_size, patch_size=s.patch_size,
                           projection_dim=s.projection_dim, hidden_act=s.hidden_act, layer_norm_eps=s.layer_norm_eps)
    hip, ref = _pair(cfg)
    px = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(9)).half()
    got = encode_image_embeddings(hip, px.to(DEV))
    torch.cuda.synchronize()
    with torch.no_grad():
        want = ref(px.float()).image_embeds.unsqueeze(1)
    assert got.shape == (1, 1, 1024) and torch.isfinite(got).all()
    err = rel_l2(got.float().cpu(), want)
    assert err <= 1e-2, f"ViT-H/14 image_embeds rel_l2={err:.3e}"


def test_clip_argument_checks():
    from transformers import CLIPVisionConfig
    cfg = CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2, image_size=28,
                           patch_size=14, projection_dim=64)
    hip, _ = _pair(cfg)
    with pytest.raises(ValueError):
        hip(torch.zeros(1, 3, 56, 56, dtype=torch.float16, device=DEV))
    with pytest.raises(TypeError):
        hip(torch.zeros(1, 3, 28, 28, dtype=torch.float32, device=DEV))

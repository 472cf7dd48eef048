"""Randomised cross-check of sp_gemm_f16 (tools/fuzz_gemm.py): random shapes / gather modes / epilogue flags against fp32
torch on the CPU, with guard rows and columns around the output (no write outside [0, m) x [0, n_store)).
Tolerance 3e-3 relative L2 per case (fp16 storage, fp32 accumulation), every kernel family."""
import os
import random
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("route,bm", [(0, 0), (2, 256), (2, 192), (2, 128), (1, 0), (3, 256), (3, 192), (3, 128), (3, -192), (4, 0)])
def test_gemm_fuzz(route, bm):
    """route: sp_gemm_set_route (0 automatic, 1 small tiles, 2 ping-pong, 3 persistent-stream, 4 split-K); route 3 draws
    long-K, many-row linear shapes more often so that workgroups walk several tiles, route 4 few-row shapes with
    N % 256 == 0 and hands every call a workspace (guarded)."""
    import fuzz_gemm
    from vdpp_amd.hip import ops
    seed = 1000 + bm + route
    rng, g = random.Random(seed), torch.Generator().manual_seed(seed)
    # bm -192: the 256 x 192 persistent tiles (widths that are multiples of 192); other shapes take the automatic choice
    with (ops.gemm_route(3, bm=256, bn=192) if bm == -192 else ops.gemm_route(route, bm=bm)):
        worst = max(fuzz_gemm.one(rng, g, stream_shapes=route == 3, splitk=route == 4) for _ in range(30))
    assert worst <= 3e-3


def test_attention_and_norm_fuzz():
    """tools/fuzz_misc.py: random shapes of the spatial (fp16 and fp8) / temporal attention, GroupNorm (three-launch
    and single-launch paths), LayerNorm, and (round 3) the CLIP small-sequence attention, the VAE row softmax and GELU
    kernels against fp32 torch, with guard rows around every output.
    Tolerances: 3e-3 relative L2 (fp16 kernels), 3e-2 for the fp8 attention against the e4m3-rounded inputs."""
    import fuzz_misc
    rng, g = random.Random(77), torch.Generator().manual_seed(77)
    for _ in range(90):
        fuzz_misc.one(rng, g)

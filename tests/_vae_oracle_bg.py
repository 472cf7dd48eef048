"""The host-side half of tests/test_vae_gpu.py::test_decoder_benchmark_shape_matches_oracle, as a process of its own: the
fp32 oracle decode of the demo's 14 x 72 x 128 latent (97 TFLOP, minutes on the box's host cores).  tests/conftest.py starts
it when the GPU session begins, so it runs BESIDE the other GPU tests instead of in front of the suite's time limit; the
test collects the result.  CPU only: this process never touches the GPU (it does not count against the box's limit of
processes on the card).  usage: _vae_oracle_bg.py OUT.pt THREADS"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def inputs():
    """(state_dict, z): the weights and the latent chunk of the test, from fixed seeds (the test builds the same)."""
    import torch
    import vdpp_amd  # noqa: F401
    from vdpp_amd.models.vae_hip import VAEDecoderConfig, random_state_dict

    sd = random_state_dict(VAEDecoderConfig.svd(), seed=3)
    g = torch.Generator().manual_seed(31)
    z = (torch.randn(14, 4, 72, 128, generator=g) * 4.0).half()
    return sd, z


def main(out, threads):
    os.environ["HIP_VISIBLE_DEVICES"] = ""          # belt and braces: nothing here may open the card
    import torch
    from oracle.vae_temporal_decoder_ref import TemporalDecoderRef
    from oracle.vae_temporal_decoder_ref import VAEDecoderConfig as RefCfg

    torch.set_num_threads(threads)
    sd, z = inputs()
    ref = TemporalDecoderRef(RefCfg.svd()).eval()
    ref.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    with torch.no_grad():
        want = ref(z.float(), 14)
    torch.save(want.half(), out + ".tmp")           # fp16 on disk: 0.5 GB instead of 1 (the comparison is at 2e-2)
    os.replace(out + ".tmp", out)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))

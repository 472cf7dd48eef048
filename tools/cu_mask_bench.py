#!/usr/bin/env python3
"""Two videos in flight, each on its own HIP stream: does giving each stream its own half of the chip (CU masks:
hipExtStreamCreateWithCUMask) beat letting both streams' kernels compete for all 256 CUs?  HBM-bound kernels of one
video would then always run beside the MFMA-bound ones of the other.  usage: cu_mask_bench.py [pattern ...]
patterns: none | halves (bits 0..127 / 128..255) | xcd (bit i -> stream (i % 8) // 4) | pairs (bit i -> (i // 2) % 2)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vdpp_amd  # noqa
from vdpp_amd.models.svd_unet import StableVideoUNet

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (w * 32 + b) in bits) for w in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def streams_for(pattern):
    if pattern == "none":
        return [torch.cuda.Stream(), torch.cuda.Stream()]
    pick = {"halves": lambda i: i // 128, "xcd": lambda i: (i % 8) // 4, "pairs": lambda i: (i // 2) % 2,
            "xcdblock": lambda i: (i // 32) // 4 % 2}[pattern]
    return [masked_stream({i for i in range(256) if pick(i) == s}) for s in range(2)]


def main():
    dev = torch.device("cuda:0")
    steps = int(os.environ.get("STEPS", 10))
    model = StableVideoUNet.from_random_init(StableVideoUNet._default_timestep_schedule(25), seed=0, device=dev)
    model.set_dummy_conditioning(1, 14, 72, 128, dev)
    lat0 = torch.randn(1, 4, 14, 72, 128, device=dev, dtype=torch.float16) * model.init_noise_sigma
    for pattern in (sys.argv[1:] or ["none", "halves", "xcd"]):
        sts = streams_for(pattern)
        best = 1e9
        for rep in range(3):
            lats = [lat0.clone(), lat0.clone()]
            torch.cuda.synchronize()
            for s in sts: s.wait_stream(torch.cuda.current_stream())
            t0 = time.perf_counter()
            with torch.no_grad():
                for step in range(steps):
                    for j in range(2):
                        with torch.cuda.stream(sts[j]):
                            lats[j] = model(lats[j], step)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"{pattern:9s}: {1e3 * best / (2 * steps):7.2f} ms per UNet forward (two videos in flight, best of 3)", flush=True)


if __name__ == "__main__":
    main()

"""The C-ABI library loads without a GPU and exports every symbol include/svdpipe.h declares; argument
validation rejects bad descriptors before anything is launched (no compute calls here)."""

import ctypes
import os
import re

import pytest

from vdpp_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    header = open(os.path.join(ROOT, "include", "svdpipe.h")).read()
    declared = set(re.findall(r"\b(sp_[a-z0-9_]+)\s*\(", header))
    declared -= {"sp_gemm_desc"}
    assert declared, "no declarations parsed"
    lib = hip.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in svdpipe.h but not exported"
    assert declared == set(hip.SIGNATURES), "binding table and header disagree"
    assert lib.sp_version() >= 100


def test_gemm_desc_layout_matches_header():
    # field order of the ctypes mirror == field order in the header
    header = open(os.path.join(ROOT, "include", "svdpipe.h")).read()
    body = header[header.index("typedef struct sp_gemm_desc {"):header.index("} sp_gemm_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for stmt in body.split("{", 1)[1].split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        parts = stmt.split(",")
        first = re.sub(r"^(const\s+)?(void|float|int64_t|size_t|int)\s*\**\s*", "", parts[0].strip())
        names.append(first.strip(" *"))
        names.extend(p.strip(" *") for p in parts[1:])
    assert names == [f[0] for f in hip.GemmDesc._fields_]


def test_rejects_bad_arguments_without_launching():
    lib = hip.load()
    d = hip.GemmDesc()
    assert lib.sp_gemm_f16(ctypes.byref(d), None) == -1
    assert b"null" in lib.sp_last_error()
    assert lib.sp_groupnorm_f16(None, None, None, None, 1, 1, 8, 32, 1e-5, 0, None, 0, None) == -1
    assert lib.sp_layernorm_f16(None, None, 0, None, None, None, None, 1, 8, 1e-5, None) == -1
    assert lib.sp_attn_spatial_f16(None, None, None, None, 64, 64, 64, 64, 1, 1, 1, 0.125, None, None) == -1
    assert lib.sp_attn_spatial_fp8(None, None, None, None, 64, 64, 64, 64, 1, 1, 1, 0.125, None, 0, None, None) == -1
    assert lib.sp_attn_fp8_ws_bytes(14, 9216, 5) == 3 * 14 * 9216 * 5 * 64
    assert lib.sp_dummy_unet_f32(*([None] * 9), 1e-5, 1, 0.5, 1, 8, 16, 1, 1, 1, None) == -1
    assert lib.sp_groupnorm_ws_bytes(14, 9216, 320, 32) > 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        hip.load()


def test_documented_binding_in_integration_md_matches_the_library():
    """INTEGRATION.md is where a reference maintainer copies the ctypes stub from: its GemmDesc must be the library's
    struct field for field (a shorter mirror makes sp_gemm_f16 read past the caller's struct), and it must carry the
    load-time size check."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    start = text.index("class GemmDesc(ctypes.Structure)")
    block = text[start:text.index("lib.sp_gemm_f16.argtypes", start)]
    fields_src = block[block.index("_fields_ = ["):]
    pairs = re.findall(r'\("([a-z0-9_]+)",\s*ctypes\.(c_[a-z0-9_]+)\)', fields_src)
    assert pairs, "no fields parsed out of INTEGRATION.md"
    mirror = [(name, getattr(ctypes, ctype)) for name, ctype in pairs]
    assert mirror == list(hip.GemmDesc._fields_)
    doc_struct = type("DocGemmDesc", (ctypes.Structure,), {"_fields_": mirror})
    assert ctypes.sizeof(doc_struct) == hip.load().sp_gemm_desc_size()
    assert "sp_gemm_desc_size()" in block and "raise" in block, "the documented stub must refuse a size mismatch"


def test_clock_stamps_pair_first_and_last_stamp_per_xcd():
    """ops.ClockStamps.ghz() (host logic of bench.py's `roofline.clock_ghz_live`): the first and the last stamp are paired XCD by
    XCD -- the shader-clock counters of different XCDs need not share an origin -- and the clock is 0.1 GHz x shader ticks per
    100 MHz tick; fewer than two stamps give None.  No GPU: the stamp records are written by hand."""
    import torch
    from vdpp_amd.hip import ops
    st = ops.ClockStamps(torch.device("cpu"), 3)
    assert st.ghz() is None
    origin = {0: 10**9, 1: 5 * 10**12, 5: 77}               # per-XCD origins of the shader-clock counter
    def write(slot, real, ghz):
        for b in range(st.BLOCKS):
            x = (0, 1, 5)[b % 3]
            st.buf[slot, b, 0] = x
            st.buf[slot, b, 1] = origin[x] + int(real * ghz * 10)     # shader ticks at `ghz` since real tick 0
            st.buf[slot, b, 2] = 4000 + real + (b % 2)                # workgroups of one stamp run within a tick or two
            st.buf[slot, b, 3] = 1
    write(1, 0, 1.8)                 # slots in any order: ghz() orders stamps by their 100 MHz time
    write(0, 2_000_000, 1.8)         # 20 ms later
    write(2, 1_000_000, 1.8)
    st.n = 3
    ghz, secs, xcds = st.ghz()
    assert xcds == 3
    assert abs(ghz - 1.8) < 1e-3
    assert abs(secs - 0.02) < 1e-4
    st.n = 1
    assert st.ghz() is None

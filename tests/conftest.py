import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import vdpp_amd  # noqa: E402,F401  (registers the hyphenated package directory as `vdpp_amd`)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


BG_ORACLE = {}      # name -> (Popen, output path) of host-side oracle runs started beside the GPU session


def pytest_collection_finish(session):
    """The longest host-side oracle of the GPU suite (the fp32 temporal-VAE decode at the demo's size, ~3 minutes of the
    box's host cores) starts NOW, as a CPU-only process, and runs beside the other GPU tests; its test collects the
    result at the end (tests/_vae_oracle_bg.py)."""
    import subprocess
    import tempfile

    import torch

    wanted = [it for it in session.items if it.name == "test_decoder_benchmark_shape_matches_oracle"
              and not any(m.name == "skip" for m in it.iter_markers())]
    if not wanted or not torch.cuda.is_available() or session.config.option.collectonly:
        return
    out = os.path.join(tempfile.gettempdir(), f"vae_oracle_{os.getpid()}.pt")
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 2)
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_vae_oracle_bg.py"), out, str(max(2, cores // 2))],
                            env=dict(os.environ, HIP_VISIBLE_DEVICES="", PYTHONDONTWRITEBYTECODE="1"))
    BG_ORACLE["vae_benchmark_shape"] = (proc, out)


def pytest_sessionfinish(session, exitstatus):
    for proc, out in BG_ORACLE.values():
        if proc.poll() is None:
            proc.kill()
        for f in (out, out + ".tmp"):
            if os.path.exists(f):
                os.remove(f)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

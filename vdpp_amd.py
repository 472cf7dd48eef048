"""Import alias for the ``video-diffusion-pipeline-parallel_amd`` package.

The on-disk package directory carries the upstream project's hyphenated name, which
is not a legal Python identifier.  Importing this module registers that directory in
``sys.modules`` under the importable name ``vdpp_amd`` so that
``import vdpp_amd.pipeline`` / ``from vdpp_amd.models import StableVideoUNet`` work.
"""

from __future__ import annotations

import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "video-diffusion-pipeline-parallel_amd")


def _load() -> None:
    spec = importlib.util.spec_from_file_location(
        "vdpp_amd",
        os.path.join(_PKG_DIR, "__init__.py"),
        submodule_search_locations=[_PKG_DIR],
    )
    module = importlib.util.module_from_spec(spec)
    sys.modules["vdpp_amd"] = module
    spec.loader.exec_module(module)


_load()

import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` on a host WITHOUT an MI355X (this build container): the GPU tests are skipped with a reason instead of failing one by
    one inside the library.  The product itself never falls back: a missing device raises (alignsdf_amd/_native.py,
    HipSdfDecoder.__init__).  ASDF_REQUIRE_GPU=1 turns the skip back into failures (a GPU box whose device has gone away)."""
    if os.environ.get("ASDF_REQUIRE_GPU"):
        return
    gpu_items = [it for it in items if it.get_closest_marker("gpu") is not None]
    if not gpu_items:
        return
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if not have:
        skip = pytest.mark.skip(reason="needs an MI355X (torch.cuda.is_available() is False on this host)")
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def native_lib():
    """The in-tree HIP library; built on demand (hipcc cross-compiles without a GPU)."""
    from alignsdf_amd import _native, build_native
    if not os.path.exists(_native.LIB_PATH):
        build_native.build()
    return _native.lib()

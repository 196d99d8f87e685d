import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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

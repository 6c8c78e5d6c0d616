import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library must exist for every test tier (CPU tests load it, GPU tests run it)."""
    import rag_uq_amd  # noqa: F401
    from rag_uq_amd import _native
    if not _native.LIB_PATH.exists():
        import __graft_entry__
        __graft_entry__.build()
    return _native.load_library()

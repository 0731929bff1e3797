import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The built libraries are git-ignored; they normally travel with the tree.  On a checkout without them, build once
    # (hipcc cross-compiles without a GPU; gcc builds the C oracle) instead of failing every test at import.
    lib = os.path.join(ROOT, "cartpolesimulation_amd", "libcpmppi.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

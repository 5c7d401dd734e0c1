import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fea-large_amd"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

DECKS = os.path.join(ROOT, "tests", "golden", "decks")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def decks_dir():
    return DECKS


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Everything under test is native code: make sure it is built."""
    need = [os.path.join(ROOT, "fea-large_amd", "libfeahip.so"),
            os.path.join(ROOT, "fea-large_amd", "libfeahost.so"),
            os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()

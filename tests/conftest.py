import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _libemx_is_built():
    """The .so is git-ignored: build it in-tree when it is missing or older than its sources
    (hipcc cross-compiles gfx950 without a GPU).  No fallback: a failed build fails the session."""
    from emcee_amd import _build
    if _build.stale():
        _build.build()
    yield

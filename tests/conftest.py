import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Incremental build (hipcc needs no GPU; a fresh clone takes ~2 min, an up-to-date tree nothing): the tests
    always run the library built from the sources in the tree, never a stale or hand-copied .so."""
    from nabo_amd import _build
    if not os.environ.get("NABO_KNN_SO"):
        _build.build()


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(scope="session")
def gpu_lib():
    """The HIP library, loaded; GPU tests FAIL (not skip) when it is missing or sees no device."""
    import nabo_amd
    from nabo_amd import _lib
    _lib.lib()
    assert nabo_amd.device_count() > 0, "no HIP device visible: -m gpu tests need an MI355X"
    return nabo_amd

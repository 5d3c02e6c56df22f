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


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def load_golden():
    return golden


@pytest.fixture(scope="session", autouse=True)
def _libpbl_built():
    """Every test needs libpbl.so (the packer lives in it): build it once if it is missing or stale (hipcc cross-compiles
    gfx950 without a GPU; the build is lock-protected, so xdist workers do not race)."""
    import __graft_entry__ as g
    g.build()

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a checkout without the in-tree build (the .so is git-ignored): compile it once, here or on the GPU box
    from genvc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        from genvc_amd.build import build
        build(verbose=False)


def golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.fixture(scope="session")
def gold():
    return golden

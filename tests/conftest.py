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
    # the oracle's batch-1 GEMVs collapse beyond a few dozen threads (bench.py's sweep on the GPU box's 256-thread host: 0.2 s on 16 threads,
    # 1.0 s on 64, 150 s on 256 for the same three steps): cap torch's intra-op pool for every oracle-driven test
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 8))
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

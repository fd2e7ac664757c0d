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


@pytest.fixture(scope="session")
def ora():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def fish(golden_dir):
    import numpy as np
    # test/test.cpp:73,85: 8-bit gray JPEG -> Mat1f, unscaled 0..255
    return np.load(os.path.join(golden_dir, "fish_u8.npy")).astype(np.float32)

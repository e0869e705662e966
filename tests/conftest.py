import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): oracle/rm_oracle.c through ctypes."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def rm():
    """The product package; building it needs hipcc only (cross-compiles without a GPU)."""
    from cpu_raymarcher_amd import _native
    _native.build()
    import cpu_raymarcher_amd as R
    return R


@pytest.fixture(scope="session")
def gpu_ctx(rm):
    return rm.Context(0)


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_crops():
    import numpy as np
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_crops.npz")))

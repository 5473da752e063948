import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_helpers.npz"))


@pytest.fixture(scope="session")
def hip_lib():
    """Builds (if needed) and loads the product library; used by CPU tests that only inspect symbols."""
    import subprocess
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "das3r_amd", "csrc")])
    from das3r_amd import _lib
    return _lib.load()

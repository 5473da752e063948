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


@pytest.fixture(autouse=True)
def _switches_follow_env(monkeypatch):
    """The library reads its DAS3R_* experiment switches once and again on das3r_reload_switches() (api.hip).  Tests flip them
    through monkeypatch.setenv / delenv: forward those to the library when it is loaded, and start every test from the
    environment as it is (the previous test's changes have been undone by then)."""
    from das3r_amd import _lib

    def reload():
        if _lib._lib is not None:
            _lib.reload_switches()

    reload()
    set0, del0 = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, prepend=None):
        set0(name, value, prepend)
        if name.startswith("DAS3R_"):
            reload()

    def delenv(name, raising=True):
        del0(name, raising)
        if name.startswith("DAS3R_"):
            reload()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield

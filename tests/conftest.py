"""pytest configuration: the `gpu` marker (tests that need a real MI355X) and shared fixtures."""
import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def refmex():
    """The compiled reference MEX (oracle/_ref).  Built on demand when /root/reference is present."""
    from oracle import refmex as rm
    if not rm.available():
        import subprocess
        if os.path.isdir("/root/reference"):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not rm.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return rm.RefMex()


@pytest.fixture(scope="session")
def glue(refmex):
    from oracle import glue as g
    return g.Glue(refmex)

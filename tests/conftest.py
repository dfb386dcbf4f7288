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
    from oracle.oracle import Oracle
    return Oracle(threads=1)


@pytest.fixture(scope="session")
def reference():
    from oracle.oracle import Reference
    if not Reference.available() and not os.path.isdir("/root/reference"):
        pytest.skip("compiled reference (oracle/_ref) not available here")
    return Reference()


@pytest.fixture(scope="session")
def ctx():
    """A libmgm_hip context on device 0 (GPU tests only).  No CPU fallback exists."""
    import mgm_amd
    c = mgm_amd.Context(0)
    yield c
    c.close()

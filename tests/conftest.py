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


def pytest_collection_modifyitems(config, items):
    """Everything in tests/test_gpu_*.py needs the device: mark it `gpu` even if a file forgets its pytestmark."""
    for item in items:
        if os.path.basename(str(item.fspath)).startswith("test_gpu_"):
            item.add_marker(pytest.mark.gpu)


@pytest.fixture(scope="session")
def ctx():
    """A libmgm_hip context on device 0 (GPU tests only).  No CPU fallback exists: without a device the
    tests that need one are skipped, not run on something else."""
    import mgm_amd
    try:
        c = mgm_amd.Context(0)
    except mgm_amd.MgmError as e:
        if e.code == mgm_amd.MGM_ERR_HIP:
            pytest.skip("no MI355X in this box (mgm_ctx_create: MGM_ERR_HIP)")
        raise
    yield c
    c.close()

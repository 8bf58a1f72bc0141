import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "svt-av1-mod-by-patman_amd"), os.path.dirname(__file__)):
    if p not in sys.path:
        sys.path.insert(0, p)


# the failure-injection hooks of the library are inert without this (include/svt_hip.h "test hooks"); set before the library loads
os.environ.setdefault("SVTAV1_HIP_TEST_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import pyorc
    return pyorc.oracle()


@pytest.fixture(scope="session")
def ref():
    import pyorc
    if not pyorc.have_ref():
        pytest.skip("oracle/_ref/libsvtref.so not built (needs /root/reference; built by __graft_entry__.build())")
    return pyorc.ref()


@pytest.fixture(scope="session")
def hip():
    """The HIP library on a real GPU.  Fails (does not skip) if the extension is missing."""
    from svtav1_hip import abi
    lib = abi.load()
    rc = lib.svt_hip_init(0)
    assert rc == 0, lib.svt_hip_last_error().decode()
    return lib

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reflib():
    from oracle.oracle import RefLib, build
    if not RefLib.available() and os.path.isdir(os.environ.get("LWS_REFERENCE", "/root/reference")):
        build()   # the build container: compile the reference in place (oracle/Makefile); a failure here is a failure
        assert RefLib.available(), "oracle/_ref/liblws_ref.so could not be built although the reference tree is mounted"
    if not RefLib.available():
        pytest.skip("oracle/_ref/liblws_ref.so absent and no reference tree to build it from (GPU box)")
    return RefLib()


def have_gpu():
    try:
        import lws_amd
        return lws_amd._capi.load().lws_device_count() > 0
    except OSError:
        return False

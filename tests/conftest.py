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
    from oracle.oracle import RefLib
    if not RefLib.available():
        pytest.skip("oracle/_ref/liblws_ref.so not built (reference tree absent)")
    return RefLib()


def have_gpu():
    try:
        import lws_amd
        return lws_amd._capi.load().lws_device_count() > 0
    except OSError:
        return False

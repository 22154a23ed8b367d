import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "e-d3dgs_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import pytest


@pytest.fixture
def libopt():
    """Set process-wide switches of the HIP library for one test (ed3dgs_set_option; the library reads the environment only
    when it is loaded) and restore them afterwards: libopt("DEFORM_FP32_MFMA", 1)."""
    from ed3dgs_amd import _lib
    old = {}

    def set_(name, value=1):
        prev = _lib.set_option(name, value)
        old.setdefault(name, prev)

    yield set_
    for k, v in old.items():
        _lib.set_option(k, v)

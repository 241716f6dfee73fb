import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_device():
    """Index of a usable HIP device; GPU tests FAIL (not skip) when the native path is unusable."""
    from tscm_calib_amd import lib
    n = lib.lib().tscm_device_count()
    assert n > 0, "no HIP device visible: -m gpu tests need the MI355X"
    return 0

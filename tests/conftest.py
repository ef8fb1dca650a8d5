import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu() -> bool:
    try:
        from eesen_amd import _lib
        return _lib.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must FAIL (not skip) when the HIP library is missing or no device is visible."""
    from eesen_amd import _lib
    lib = _lib.load()
    n = _lib.device_count()
    assert n > 0, "no HIP device visible: -m gpu tests need an MI355X"
    return lib

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """HIP context on cuda:0 -- fails loudly (no CPU fallback) when the library or the device is missing."""
    import femus_amd
    c = femus_amd.Context(0)
    yield c
    c.close()


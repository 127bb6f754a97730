import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def af():
    """The product package (host mirror over the C ABI)."""
    import audio_flow_rs_amd
    return audio_flow_rs_amd


@pytest.fixture(scope="session")
def po():
    """The CPU oracle (checker)."""
    from oracle import pyoracle
    return pyoracle

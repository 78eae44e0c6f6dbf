import os
import sys

import pytest

try:
    # torch first: it ships its own HIP runtime, and a process must not end up with two of them
    # (loading libveloslam_amd.so before torch leaves torch without a device: "No HIP GPUs")
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/liboracle.so) -- the checker, never the product."""
    from oracle import oracle as orc
    orc.lib()
    return orc

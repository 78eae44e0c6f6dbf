import os
import sys

import pytest

try:
    # torch first: it ships its own HIP runtime, and a process must not end up with two of them
    # (loading libveloslam_amd.so before torch leaves torch without a device: "No HIP GPUs")
    import torch  # noqa: F401
except ImportError:
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The hot path first.  `pytest -x` stops at the first failure, and in round 3 a race in a multi-GPU *plumbing*
# test (alphabetically early) kept all of the K1/K2/K3 parity tests from running on the driver's box.  GPU tests
# run in this order; inside a file the written order stays.  Files not listed go last, CPU tests are untouched
# (they keep their alphabetical order in front / between, whatever `-m` selects).
_GPU_ORDER = ["test_gpu_parity", "test_gpu_mapping", "test_golden_icp", "test_gpu_fuzz", "test_gpu_hash", "test_gpu_knn",
              "test_gpu_batch_invariance", "test_gpu_sum_definition", "test_gpu_streams", "test_drive", "test_cpp_api", "test_gpu_comm", "test_bench_cli"]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        if item.get_closest_marker("gpu") is None:
            return -1
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _GPU_ORDER.index(name) if name in _GPU_ORDER else len(_GPU_ORDER)

    items.sort(key=rank)      # stable: CPU tests first in collection order, then the GPU files as listed


@pytest.fixture(scope="session", autouse=True)
def _order_torch_producers_before_ctx_calls():
    """include/velo.h STREAM CONTRACT: a ctx runs on its own non-blocking stream, so a tensor a test made on
    torch's stream (torch.full, .cuda(), indexing) must be complete before its pointer is handed to a *_dev
    call.  Tests say so where they do it (torch.cuda.synchronize()); this hook is the safety net for the one
    that forgets -- it waits for TORCH'S stream only, never for the ctx's own streams, so the library's
    internal ordering is still what is under test."""
    from veloslam_amd import capi
    if torch is not None and torch.cuda.is_available():
        capi.set_producer_sync(lambda: torch.cuda.current_stream().synchronize())
    yield
    capi.set_producer_sync(None)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/liboracle.so) -- the checker, never the product."""
    from oracle import oracle as orc
    orc.lib()
    return orc

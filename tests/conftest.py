import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def gpu_ctx():
    # PyTorch brings its own copy of the HIP runtime: when both live in one
    # process torch has to initialise first (bench.py does the same).
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    torch.cuda.init()
    from sfm_amd import matcher
    ctx = matcher.Context(0)
    yield ctx
    ctx.close()


_poison_count = [0]


@pytest.fixture(autouse=True)
def _poison_lds_before_gpu_tests(request):
    """Every GPU test starts with the LDS of every CU overwritten: a kernel that
    skips a DMA or reads a stage too early must not find the right bytes there
    because the previous test ran on the same data."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    ctx = request.getfixturevalue("gpu_ctx")
    _poison_count[0] += 1
    ctx.check(ctx._L.sfm_gpu_debug_poison_lds(ctx.handle, 0xA5A50000 + _poison_count[0]))
    yield

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from videoglamm_amd import _lib

    lib = _lib.load()
    ncu = lib.vg_init(0)
    assert ncu > 0, lib.vg_last_error()
    return torch.device("cuda:0")


@pytest.fixture()
def cpu_ops(monkeypatch):
    """Swap the HIP operator layer for the plain-PyTorch statements (tests/_cpu_ops.py) so the host-side graph
    code can be exercised without a GPU.  Test-only: the product never falls back."""
    import _cpu_ops
    from videoglamm_amd import ops

    for name in _cpu_ops.ALL:
        if hasattr(ops, name):
            monkeypatch.setattr(ops, name, getattr(_cpu_ops, name))
    return ops

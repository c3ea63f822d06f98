import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_ops_factory():
    import torch
    from open_pandora_amd.ops_hip import HipOps

    cache = {}

    def make(dtype):
        if dtype not in cache:
            cache[dtype] = HipOps(dtype=dtype, device="cuda:0")
        return cache[dtype]

    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    return make

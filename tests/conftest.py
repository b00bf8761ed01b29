import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running convergence test")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def cuda():
    import torch

    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _seed_global_generators():
    """Every test starts from the same state of the tensor library's global generators (CPU and GPU), so a test
    that draws from them sees the same numbers whichever tests ran before it."""
    import torch

    torch.manual_seed(1234)
    yield

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Before anything loads HIP: with eight hardware queues the library keeps two frames in flight by
# default, which is the configuration the GPU tests are meant to exercise (and the one to deploy).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_available():
    import torch
    return torch.cuda.is_available()

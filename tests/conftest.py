import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# before anything loads an OpenMP runtime (torch, the oracle): keep it within the job's CPU share
from oracle_lib import limit_openmp_threads  # noqa: E402
limit_openmp_threads()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests must never silently pass on a box without a GPU: skip them loudly here,
    # they are selected with `-m gpu` on the GPU box.
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)

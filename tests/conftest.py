import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def host_threads():
    """All host cores for one big CPU-oracle run, restored afterwards: hundreds of OpenMP threads left on make the small
    ops of the following oracle runs (VAE tiles) crawl for minutes."""
    import torch
    before = torch.get_num_threads()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, 64)))
    yield
    torch.set_num_threads(before)

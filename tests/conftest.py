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


def measured(name: str, value: float, bar: float) -> float:
    """An assert that can fail for a real regression: `bar` is pinned at about twice the value measured on the MI355X (the
    measured value stands beside each call).  With APEX_RECORD_MEASURED=<file> every call appends {name, value, bar} to that
    JSON-lines file, so a GPU run prints what the bars should be re-pinned to."""
    import json
    value = float(value)
    path = os.environ.get("APEX_RECORD_MEASURED")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps({"name": name, "value": value, "bar": bar}) + "\n")
    assert value < bar, f"{name}: measured {value:.3e} is not under its bar {bar:.1e}"
    return value

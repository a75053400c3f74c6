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


def pytest_sessionstart(session):
    """The CPU oracle runs through torch's intra-op thread pool, whose default is the number of cores the MACHINE has; a GPU box
    hands a job a 16-core share of 128, and 128 threads on 16 cores run the oracle several times SLOWER than 16 (bench.py's
    cpu_baseline found the same).  Cap the pool at 16 (fewer where the machine has fewer)."""
    try:
        import torch
        torch.set_num_threads(max(1, min(torch.get_num_threads(), 16)))
    except Exception:
        pass

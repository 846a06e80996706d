import os
import sys

import pytest

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # (mdie_amd/__init__.py: the product's runtime default, set before anything touches the GPU)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The GPU box reports every host core (os.cpu_count() = 256) but grants a 16-core share: torch's default thread pool
    # then oversubscribes 16x and every CPU oracle call crawls.  Cap the pool at what is actually available.
    try:
        import torch
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        torch.set_num_threads(max(1, min(16, avail)))
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

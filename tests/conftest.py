import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _release_gpu_memory_between_tests(request):
    """A training arena is tens of GB (52 GiB at 512x512 B=4); trainers of finished tests are garbage but sit in reference cycles and in
    the caching allocator's reserve until someone collects them.  Without this the suite ran the 288 GB card out of memory near its end
    (profiles/r03_gpu_tests.log, round 3)."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        try:
            import torch
            if torch.cuda.is_available() and torch.cuda.memory_reserved() > (64 << 30):      # only when it matters: the sweep costs ~0.4 s
                import gc
                gc.collect()
                torch.cuda.empty_cache()
        except Exception:
            pass

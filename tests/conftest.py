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


def _sweep_gpu_memory(min_reserved_gib):
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.memory_reserved() > (min_reserved_gib << 30):
            import gc
            gc.collect()
            torch.cuda.empty_cache()
    except Exception:
        pass


@pytest.fixture(autouse=True, scope="module")
def _release_gpu_memory_between_modules(request):
    """A training arena is tens of GB (52 GiB at 512x512 B=4); trainers of finished tests are garbage but sit in reference cycles and in
    the caching allocator's reserve until someone collects them.  Without this the suite ran the 288 GB card out of memory near its end
    (round 3: test_step_gradients_512_b4 could not get its arena with 194 GiB still reserved by finished tests).  One sweep per module
    (a sweep costs ~0.4 s: after every test it added three minutes to the suite); per test only as an emergency brake."""
    yield
    _sweep_gpu_memory(8)


@pytest.fixture(autouse=True)
def _release_gpu_memory_emergency(request):
    yield
    if request.node.get_closest_marker("gpu") is not None:
        _sweep_gpu_memory(160)

import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "intool-rag_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests fail loudly (not skip) if the native library cannot run: no silent fallbacks."""
    if not _gpu_available():
        pytest.fail("torch.cuda.is_available() is False on a run selected with -m gpu")
    import hiprag  # noqa: F401  (raises if libhiprag.so is missing)
    return 0

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as graft  # noqa: E402

graft.load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run the native HIP path: fail loudly (never skip silently to a fallback)."""
    from fast_llama_amd import capi
    if not _gpu_available():
        pytest.skip("no GPU in this container (selected with -m gpu on the GPU box)")
    capi.lib()   # raises FlmError if the HIP library is missing
    return capi

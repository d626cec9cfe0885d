import os
import sys

import pytest

# before anything can load an OpenMP runtime (numpy does not, torch and the oracle do): see oracle_py._tame_openmp
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(8, os.cpu_count() or 1))))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
# tests/test_gpu_tp.py runs up to 8 tensor-parallel ranks as 8 streams of ONE process on ONE GPU; their flag waits need one hardware
# queue per stream (HIP's default is 4 per process: two streams sharing a queue would put a waiting kernel in front of the kernel
# it waits for).  Read once, when the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as graft  # noqa: E402

graft.load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _hip_device_count():
    """ask HIP itself (not torch): a broken torch on the GPU box must not turn the parity suite into skips"""
    import ctypes
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            hip = ctypes.CDLL(name)
        except OSError:
            continue
        n = ctypes.c_int(0)
        try:
            return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
        except Exception:
            return 0
    return 0


@pytest.fixture(scope="session")
def gpu(request):
    """GPU tests run the native HIP path or FAIL: when the run selected them (`-m gpu`, or FLM_REQUIRE_GPU=1) a missing device
    or a missing library is an error, never a silent skip to nothing."""
    from fast_llama_amd import capi
    if _hip_device_count() < 1:
        markexpr = request.config.getoption("-m") or ""
        if os.environ.get("FLM_REQUIRE_GPU") == "1" or ("gpu" in markexpr and "not gpu" not in markexpr):
            pytest.fail("GPU tests were selected but HIP reports no device")
        pytest.skip("no GPU in this container (selected with -m gpu on the GPU box)")
    capi.lib()   # raises FlmError if the HIP library is missing
    return capi

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are only meaningful with a device; skip them loudly on CPU-only hosts
    # unless explicitly selected with -m gpu (then they must run, and fail if they cannot).
    if config.getoption("-m") and "gpu" in config.getoption("-m") and "not gpu" not in config.getoption("-m"):
        return
    try:
        import ctypes
        has_gpu = os.path.exists("/dev/kfd")
    except Exception:
        has_gpu = False
    if not has_gpu:
        skip = pytest.mark.skip(reason="no GPU on this host")
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _library_is_built():
    """A fresh checkout has no libqrec_hip.so (build products are not in history): build it once, like
    ``__graft_entry__.build()`` does, when the toolchain is there.  Tests never fall back to anything else."""
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "qrec_amd", "libqrec_hip.so")
    if not os.path.exists(lib) and shutil.which("hipcc") and shutil.which("make"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "qrec_amd", "csrc"), "-j8"], check=True, capture_output=True)
    yield

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# `import foodrec_amd` refuses to load without libm2d.so (there is no CPU fallback), and the .so is not in
# git: build it before anything is collected (hipcc cross-compiles gfx950 without a GPU, ~15 s).
if not os.path.exists(os.path.join(ROOT, "foodrec_amd", "libm2d.so")):
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "foodrec_amd", "csrc"), "-j", "8"], check=True,
                   stdout=subprocess.DEVNULL)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def native_lib():
    """libm2d.so, built on demand (hipcc cross-compiles gfx950 without a GPU)."""
    from foodrec_amd import _native
    return _native.lib()

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the HIP library once, here on the CPU host —
    hipcc cross-compiles gfx950 without a GPU.  On the GPU box the prebuilt library travels with the tree and this is a
    no-op; the library is never built inside a test or a GPU call."""
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "gym_amd", "lib", "libgymhip.so")
    if not os.path.exists(lib) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "gym_amd", "csrc")], check=False, capture_output=True)


@pytest.fixture(scope="session")
def oracle():
    from tests.oracle_lib import Oracle
    return Oracle()

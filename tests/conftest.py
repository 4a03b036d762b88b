"""pytest configuration: the ``gpu`` marker and import paths.

* ``-m "not gpu"`` runs everywhere (oracle vs golden vectors, host logic, C-ABI
  symbol checks, gloo multi-process tests).
* ``-m gpu`` needs one MI355X and calls the HIP path through the C-ABI.
"""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
PKG_PARENT = ROOT / "viewport-entropy-toolkit_amd"
for p in (str(ROOT), str(PKG_PARENT)):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")
    # The built library is git-ignored: build it when a fresh checkout runs the tests before
    # __graft_entry__.build() (hipcc cross-compiles gfx950 without a GPU).
    lib = PKG_PARENT / "lib" / "libvet_hip.so"
    if not lib.exists():
        import shutil
        import subprocess
        if shutil.which("make") and (shutil.which("hipcc") or Path("/opt/rocm/bin/hipcc").exists()):
            subprocess.run(["make", "-C", str(PKG_PARENT / "csrc")], check=False, capture_output=True)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

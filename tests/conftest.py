import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no libdistaff_hip.so (built artefacts are git-ignored): build it once, exactly as
    __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU).  The tests never fall back to anything else."""
    lib = os.path.join(ROOT, "distaff_amd", "libdistaff_hip.so")
    if not os.path.exists(lib) and os.environ.get("DISTAFF_HIP_LIB") is None:
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "distaff_amd", "csrc"), "-j8"])


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/_build/liboracle.so on first use."""
    import oracle as O
    O.lib()
    return O

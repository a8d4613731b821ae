import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no shared libraries (built artefacts are git-ignored): build them once, exactly as __graft_entry__.build()
    does (hipcc cross-compiles gfx950 without a GPU).  The tests never fall back to anything else.

    The tests bind the TEST build of the library, distaff_amd/libdistaff_hip_hooks.so: the same sources as the product
    (distaff_amd/libdistaff_hip.so) compiled with -DDISTAFF_TEST_HOOKS, which additionally honours the test-only DISTAFF_* switches
    (alternative formulations the tests compare with the default ones), and contains the per-operation constraint instance and the
    calibration kernels.  What is NOT run through it: bench.py (it removes DISTAFF_TEST_HOOKS and refuses anything but the product
    library), __graft_entry__.smoke(), the plain-C hosts of examples/ and tests/test_product_library.py -- they bind the product."""
    libs = [os.path.join(ROOT, "distaff_amd", n) for n in ("libdistaff_hip.so", "libdistaff_hip_hooks.so")]
    if not all(os.path.exists(p) for p in libs) and os.environ.get("DISTAFF_HIP_LIB") is None:
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "distaff_amd", "csrc"), "-j8"])
    product_only = os.environ.get("DISTAFF_PRODUCT_ONLY") == "1"       # tests/test_product_library.py re-runs a selection of the parity tests on the product library itself
    if product_only:
        os.environ.pop("DISTAFF_TEST_HOOKS", None)           # BEFORE the import: the package fixes the library it binds when it is imported
    import distaff_amd
    if product_only:
        distaff_amd.use_product()
    else:
        distaff_amd.use_test_hooks()


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/_build/liboracle.so on first use."""
    import oracle as O
    O.lib()
    return O

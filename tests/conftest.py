import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# (tests/tools/fuzz_kat.py points the known-answer tests at fixtures it made with other seeds and sizes: SIM5_GOLDEN_DIR)
GOLDEN = os.environ.get("SIM5_GOLDEN_DIR") or os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def oracle():
    import oraclelib
    oraclelib.build_oracle()
    return oraclelib.Oracle()


@pytest.fixture(scope="session")
def capi():
    """The product library through its C-ABI.  Built on demand; never replaced by a CPU path."""
    lib = os.path.join(ROOT, "sim5_amd", "lib", "libsim5gpu.so")
    if not os.path.exists(lib):
        from sim5_amd.build import build
        build()
    import sim5_amd.capi as c
    return c

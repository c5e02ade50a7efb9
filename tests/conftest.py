import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle_lib import Reference, have_reference
    if not have_reference():
        pytest.skip("oracle/_ref/libnbody_ref.so not built (needs /root/reference; see oracle/build_ref.sh)")
    return Reference()


@pytest.fixture(scope="session")
def nbx():
    """The product package; on a GPU box a missing library is a hard failure, never a skip."""
    import nbody_amd
    if not os.path.exists(nbody_amd.LIB_PATH) or not os.path.exists(os.path.join(ROOT, "nbody_sim")):
        # a checkout without build products: compile in-tree (hipcc cross-compiles gfx950 without a GPU)
        import subprocess
        subprocess.check_call(["make", "lib", "nbody_sim"], cwd=ROOT, stdout=subprocess.DEVNULL)
    nbody_amd.load_library()
    return nbody_amd


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

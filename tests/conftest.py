import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """Builds oracle/libvicoracle.so (the CPU checker) if needed."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    from oracle import pyref
    assert pyref.have_oracle()
    return pyref


@pytest.fixture(scope="session")
def ref_available():
    from oracle import pyref
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "ref_build", "build_ref.sh")])
    return pyref.have_ref("plain") and pyref.have_ref("fixed") and pyref.have_ref("compat")

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oc():
    import __graft_entry__ as ge

    return ge.load_oracle()


@pytest.fixture(scope="session")
def mi_lib():
    """ctypes binding, library loaded (no device needed)."""
    import __graft_entry__ as ge

    mi = ge.load_binding()
    mi.lib()
    return mi


@pytest.fixture(scope="session")
def mi(mi_lib):
    """binding with the device initialised -- GPU tests only; fails loudly without a GPU."""
    mi_lib.init()
    yield mi_lib

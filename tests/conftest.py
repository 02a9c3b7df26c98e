import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """Initialised HIP library; tests using it must be marked gpu.  At session end every native object still alive
    is released in a known order (ecc_ldpc_amd._lib.close_all -- the same hook atexit runs), so nothing is left for
    finalisers to do during interpreter shutdown and the process exits the normal way."""
    import ecc_ldpc_amd as E
    E.init(0)
    yield E
    E.close_all()

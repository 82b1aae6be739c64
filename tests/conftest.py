import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/mpcvr_oracle.c), built on demand.  Test infrastructure only."""
    from oracle import oracle as O
    if not os.path.exists(O.LIB_PATH):
        O.build(ref=True)
    O.lib()
    return O


@pytest.fixture(scope="session")
def mpcvr():
    """The product package; builds libmpcvr.so with hipcc if it is missing."""
    import videorenderer_amd as V
    from videorenderer_amd import api
    if not os.path.exists(api.LIB_PATH):
        from videorenderer_amd.build import build
        build()
    api.load_library()
    return V

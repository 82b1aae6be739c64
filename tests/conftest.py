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


@pytest.fixture(autouse=True)
def _test_times(request):
    """MPCVR_TEST_TIMES=<file>: one JSON line per test with its start / end on the clocks a rocprofv3 kernel trace may be stamped in —
    tests/tools/kernel_witnesses.py joins the two into 'which test launched which kernel instantiation' (profiles/<round>/kernels_by_test.json)."""
    path = os.environ.get("MPCVR_TEST_TIMES")
    if not path:
        yield
        return
    import json
    import time
    clocks = {"mono": time.CLOCK_MONOTONIC, "boot": getattr(time, "CLOCK_BOOTTIME", time.CLOCK_MONOTONIC), "real": time.CLOCK_REALTIME}
    t0 = {k: time.clock_gettime_ns(v) for k, v in clocks.items()}
    yield
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass
    t1 = {k: time.clock_gettime_ns(v) for k, v in clocks.items()}
    with open(path, "a") as f:
        f.write(json.dumps({"test": request.node.nodeid, "t0": t0, "t1": t1}) + "\n")

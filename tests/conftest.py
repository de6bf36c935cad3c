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
    """CPU oracle (test infrastructure): builds oracle/liboracle.so with gcc on first use."""
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def gpu_ctx():
    """uavqp context on cuda:0 through the C ABI; fails loudly if the HIP library is missing."""
    import uav_motion_planning_amd as u
    assert os.path.exists(u._lib.LIB_PATH), "libuavqp.so missing on the GPU box -- build() did not run"
    ctx = u.Context(0)
    yield ctx
    ctx.close()

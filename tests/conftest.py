import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure).  Built on demand with oracle/Makefile."""
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def rt_api():
    from gpu_raytracer_amd import api
    api.load()  # raises if librt_hip.so is missing: no fallback
    return api


@pytest.fixture()
def gpu_ctx(rt_api):
    ctx = rt_api.Context()
    yield ctx
    ctx.close()

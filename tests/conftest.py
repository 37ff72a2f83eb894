import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure libsdm_hip.so and the oracle exist (cross-compiles without a GPU)."""
    import __graft_entry__ as g
    lib = os.path.join(ROOT, "superviseddescent_amd", "lib", "libsdm_hip.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        g.build()
    return True


@pytest.fixture(scope="session")
def gpu_ctx(built):
    from superviseddescent_amd import Context
    ctx = Context(0)
    yield ctx
    ctx.close()

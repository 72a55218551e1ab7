import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def zk():
    """Initialised product library on cuda:0; fails loudly (no CPU fallback) when no GPU is there."""
    import zkgl

    zkgl.init(0)
    return zkgl


@pytest.fixture(scope="session")
def oracle():
    from oracle import zko

    zko.lib()
    return zko

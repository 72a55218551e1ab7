import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "era-zkevm_circuits_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # every test that takes the `zk` fixture (an initialised GPU) is a GPU test, marked or not
    # On a real device a kernel that never returns would hold the box until the caller's limit (a strike for the pool): every GPU test gets a
    # ten-minute limit enforced from a watchdog thread (a hung HIP call never returns to Python, so the signal method could not fire); the whole
    # device suite takes about three minutes, its slowest test well under one.  Not on the emulated device (ZKGL_LIB: minutes per test there).
    limit = config.pluginmanager.hasplugin("timeout") and not os.environ.get("ZKGL_LIB")
    for item in items:
        if "zk" in getattr(item, "fixturenames", ()) and not item.get_closest_marker("gpu"):
            item.add_marker(pytest.mark.gpu)
        if limit and item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(600, method="thread"))


@pytest.fixture(scope="session")
def zk():
    """Initialised product library on cuda:0.  Without a GPU the tests that need one are skipped (the library itself has
    no CPU fallback: tests/test_abi.py asserts that zk_init fails loudly)."""
    import zkgl

    if zkgl.device_count() == 0:
        pytest.skip("no GPU visible: -m gpu tests need a real MI355X")
    zkgl.init(0)
    return zkgl


@pytest.fixture(scope="session")
def oracle():
    from oracle import zko

    zko.lib()
    return zko

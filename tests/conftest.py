import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """a test that does not come back must not take the whole run (and a GPU box) with it: ten minutes each, where the
    pytest-timeout plugin is installed"""
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(600))


def _ensure_built():
    """Builds the product library and the oracle when sources are newer (cheap no-op otherwise)."""
    import __graft_entry__ as g
    g.build()


@pytest.fixture(scope="session", autouse=True)
def built():
    _ensure_built()
    yield


@pytest.fixture(scope="session")
def ref():
    """The real reference (oracle/_ref/libojph_ref.so), if it has been built."""
    from oracle import refbind
    if not refbind.available():
        pytest.skip("oracle/_ref/libojph_ref.so not built (no /root/reference here)")
    return refbind.Ref()


@pytest.fixture(scope="session")
def refgen():
    from oracle import refbind
    if not refbind.available(generic=True):
        pytest.skip("oracle/_ref/libojph_refgen.so not built")
    return refbind.Ref(generic=True)

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout(seconds): per-test limit (pytest-timeout)")


def pytest_collection_modifyitems(config, items):
    """A stuck rendezvous or kernel must fail one test, not stall the whole run: every test gets a generous limit."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def sipx():
    """The product package (directory name has a dot, so it is loaded by path)."""
    from __graft_entry__ import load_package
    return load_package()

import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "srl-zoo_amd")
for p in (PKG, REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def cabi():
    from srlz import _cabi
    return _cabi


def pytest_collection_modifyitems(config, items):
    """Every test gets a wall-clock limit (pytest-timeout, when installed): a hung loader process or kernel must fail the
    test, not stall the run."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(420))

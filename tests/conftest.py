import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A test that waits for ever (a collective one rank never joins, a device that stopped answering) must fail with a
    traceback, not sit there until somebody kills the run: every test gets a time limit when pytest-timeout is there
    (it is in this image) and the command line did not set one."""
    if not config.pluginmanager.hasplugin("timeout") or config.getoption("timeout", None):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900 if item.get_closest_marker("gpu") else 1200))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

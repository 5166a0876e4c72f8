import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_bvc():
    """Import the product package.  Its directory name has a hyphen, so it is registered
    under the importable alias `bvc_amd` (same loader `__graft_entry__` uses)."""
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def bvc():
    return load_bvc()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")

import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg(sub: str = ""):
    """The package directory is named ``ls1-mardyn_amd`` (hyphen), so it is imported through importlib."""
    name = "ls1-mardyn_amd" + ("." + sub if sub else "")
    return importlib.import_module(name)


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()

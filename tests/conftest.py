import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fftconv():
    """The product package (directory name has hyphens, so it is imported by path)."""
    import util
    return util.load_package()


@pytest.fixture(scope="session")
def oracle():
    import util
    return util.Oracle()

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mirhi():
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def scenes(mirhi):
    return mirhi.scenes


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    oracle_binding.lib()
    return oracle_binding


@pytest.fixture(scope="session")
def device(mirhi):
    dev = mirhi.Device(0)
    yield dev
    dev.destroy()

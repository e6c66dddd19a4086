import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build them once -- hipcc cross-compiles libmirhi.so for
    gfx950 without a GPU (~1 min), gcc / g++ build the oracle, libmiresources.so and the C++ host test."""
    import glob
    pkg = os.path.join(ROOT, "renderer-rs_amd")
    have = (os.path.exists(os.path.join(pkg, "libmirhi.so")) and os.path.exists(os.path.join(pkg, "libmiresources.so"))
            and glob.glob(os.path.join(ROOT, "oracle", "*.so")))
    if not have:
        import __graft_entry__ as ge
        ge.build()


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

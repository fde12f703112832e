import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/liboracle.so), built on demand.  Test infrastructure only."""
    from tests import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def hml():
    """The product library; GPU tests only."""
    import hammlet_amd
    hammlet_amd.load_library()
    return hammlet_amd

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build the HIP library and the oracle once per session."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def heights():
    import oracle_lib
    return oracle_lib.load_terrain_heights()


@pytest.fixture(scope="session")
def textures():
    import oracle_lib
    return oracle_lib.load_textures()

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import load_oracle

    return load_oracle()


@pytest.fixture()
def oracle_backend():
    """Install the oracle-backed test backend behind pf3plat_amd's wrappers (CPU tensors)."""
    from pf3plat_amd import rasterizer
    from tests.oracle_backend import OracleBackend

    be = OracleBackend()
    old = rasterizer.set_backend(be)
    yield be
    rasterizer.set_backend(old)

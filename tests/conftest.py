import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def libs():
    """Builds (if stale) and loads the product library and the oracle."""
    from fhe_string_amd import build
    build.build_product(verbose=False)
    build.build_oracle(verbose=False)
    from fhe_string_amd import _lib
    from tests import oracle_lib
    return _lib.load(strict=False), oracle_lib.load()

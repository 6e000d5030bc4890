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
    # Load order matters when RCCL is used from this library in a process that also imports torch (the two-rank tests do,
    # for gloo): torch bundles its own ROCm runtime and RCCL.  With torch imported FIRST the library's HIP dependency
    # resolves to the runtime torch has loaded and `librccl.so.1` to torch's copy -- one runtime, as in bench.py.  The other
    # order leaves two runtimes in the process and `ncclCommInitRank` cannot see the device.  A full `pytest tests` run
    # imports torch at collection time anyway (tests/test_parallel_gloo.py); this makes partial runs behave the same.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = _lib.load(strict=False)
    return lib, oracle_lib.load()

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library, initialised on cuda:0. Fails (never skips) when the HIP path is unavailable."""
    import otezip_amd as oz
    if not os.path.exists(oz.lib_path()):
        oz.build_native()
    oz.init(0)
    yield oz
    oz.shutdown()

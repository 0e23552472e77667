import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The real reference compiled into oracle/_ref (prebuilt .so; never reads /root/reference)."""
    from oracle.pyoracle import Reference

    ref = Reference()
    if not ref.available:
        pytest.skip("oracle/_ref/libhmj_ref.so not built (run `make -C oracle ref` where the reference tree exists)")
    return ref


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")

import os
import sys

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
PKG_ROOT = os.path.join(ROOT, "cav-hoomd_amd")
for p in (ROOT, PKG_ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def ref(oracle_mod):
    return oracle_mod.RefOracle("O2")


@pytest.fixture(scope="session")
def ref_o3(oracle_mod):
    return oracle_mod.RefOracle("O3")


@pytest.fixture(scope="session")
def capi():
    from cavitymd import _capi
    _capi.build()
    _capi.load()
    return _capi


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

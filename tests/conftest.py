import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _keys(name, seed):
    import oracle_lib as ol
    p = ol.params(name)
    ks = ol.KeySet(p, seed=seed)
    return ks, ol.Ctx(ks)


@pytest.fixture(scope="session")
def toy_default():
    """default128 gadget/keyswitch shape with n=24 (fast)."""
    return _keys("toy", 3)


@pytest.fixture(scope="session")
def toy_redsec():
    """redsec_small_v2 gadget/keyswitch shape with n=20 (fast)."""
    return _keys("toy_redsec", 4)


@pytest.fixture(scope="session")
def full_default():
    return _keys("default128", 42)


@pytest.fixture(scope="session")
def full_redsec():
    return _keys("redsec_small_v2", 43)

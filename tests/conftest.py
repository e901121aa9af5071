import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One HIP context for the whole GPU session.  No skip: on a GPU box a missing device / extension must fail loudly."""
    import dre_amd as D
    c = D.Context(0)
    D.set_default_context(c)
    return c


@pytest.fixture(scope="session")
def rail371():
    import dre_amd as D
    d = D.steel_profile(371)
    L, Dm = D.initial_value(d)
    return d, L, Dm

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
    """The CPU restatement (test infrastructure only)."""
    from oracle import binding

    binding.lib()
    return binding


@pytest.fixture(scope="session")
def yk():
    """The product's host-side mirror; requires libyuki_hip.so (no fallback)."""
    from yuki_amd import core

    core.lib()
    return core


@pytest.fixture(scope="session")
def ctx(yk):
    """A HIP context on cuda:0 — GPU tests only."""
    c = yk.Context(0)
    yield c
    c.close()

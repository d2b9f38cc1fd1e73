import os
import sys

import pytest

# PyTorch ships its own HIP runtime and RCCL (torch/lib); libyuki_hip.so links the system ones under the same sonames.  One
# process holds one copy of each — whichever is loaded first — and PyTorch does not initialise on the system runtime
# ("No HIP GPUs are available"), so the tests that use torch for device buffers need it loaded BEFORE the library, whatever
# subset or order of tests runs.  (A host that brings its own HIP stack does the same: load it first.)
try:
    import torch  # noqa: F401
except ImportError:  # CPU-only checkouts without torch still run the oracle / host tests
    pass

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


@pytest.fixture(scope="session")
def cfg3_scene():
    """BASELINE configs[2]'s scene (1,024,012 triangles), generated once per session; read-only."""
    from yuki_amd import scenes

    return scenes.by_name("cfg3")

"""yuki_amd — MI355X-native wavefront Path integrator for sndels/yuki's hot path.

The compute path is libyuki_hip.so (hand-written HIP for gfx950 behind the C ABI of
include/yuki_hip.h).  This package is the thin host-side mirror of the reference's
Integrator / Sampler / Film / Camera / Scene interface plus the synthetic scene
generators; importing the stage API fails loudly if the library is missing.
"""
from . import abi, scenes  # noqa: F401


def __getattr__(name):
    # lazy so that `import yuki_amd.scenes` works before the library is built
    if name in ("core", "_ffi"):
        import importlib

        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)

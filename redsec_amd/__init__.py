"""redsec_amd -- MI355X-native gate-bootstrapping backend for REDsec's encrypted hot path.

Only what the path needs lives here:
  csrc/        HIP kernels (gfx950) + the C ABI declared in include/redsec_hip.h
  backend.py   ctypes binding of that ABI (torch tensors in, torch tensors out)
  build.py     in-tree build of libredsec_hip.so (and the test-only lane emulator)
"""
from .backend import Backend, RedsecHipError, params, load_library, GATES, ABI_SYMBOLS  # noqa: F401

__all__ = ["Backend", "RedsecHipError", "params", "load_library", "GATES", "ABI_SYMBOLS"]

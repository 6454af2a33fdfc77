"""MI355X-native per-frame hot path of multi-view motion capture.

cross-view association -> multi-view DLT triangulation -> temporal IK, as
hand-written gfx950 HIP kernels behind a C ABI (include/mvmc.h), with Python
modules that keep the reference's call surface.
"""
from . import _cabi  # noqa: F401

__all__ = ["_cabi"]

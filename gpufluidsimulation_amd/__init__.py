"""MI355X-native bimocq3D hot path: HIP kernels behind the reference's gpu_* C-ABI.

The product is the C/C++ side (csrc/, include/); this package is the thin Python host mirror
used by the tests, bench.py and the smoke entry.  No CPU fallback exists on purpose.
"""
from . import _lib                      # noqa: F401
from ._lib import BimocqError, BimocqLibraryMissing, check, hip_lib   # noqa: F401
from .mapper import DeviceBuffer, GpuMapper                            # noqa: F401

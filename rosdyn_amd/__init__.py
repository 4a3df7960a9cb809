"""rosdyn_amd -- MI355X-native batched rigid-body dynamics behind the rosdyn::Chain call surface.

The product is the C-ABI shared library ``librdyn_hip.so`` (include/rdyn.h, sources in rosdyn_amd/csrc);
this package is the thin Python/ctypes mirror used by the tests and bench.py.
"""
from .chain import Chain, createChain  # noqa: F401
from ._lib import LIB_PATH, RdynError  # noqa: F401

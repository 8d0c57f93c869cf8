"""marlin_amd -- MI355X-native shard-query path for Marlin.

Host code is C (marlin_amd/host), kernels are HIP for gfx950 (marlin_amd/csrc) behind the C ABI
in include/marlin_gpu.h; this package is the ctypes plumbing used by tests and bench.py.
"""
from . import ffi  # noqa: F401
from .index import Index, ParsedQuery, QueryError  # noqa: F401
from .engine import GpuEngine, GpuError, assemble  # noqa: F401

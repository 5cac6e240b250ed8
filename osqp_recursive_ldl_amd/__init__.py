"""MI355X-native batched direct KKT backend for OSQP's ADMM loop (gfx950 HIP kernels behind the
reference's linsys_solver plugin API).  See include/osqp_rldl_hip.h for the C-ABI this package wraps.

Importing the package loads libosqp_rldl_hip.so and fails loudly if it has not been built.
"""
from . import _lib
from ._lib import build, lib  # noqa: F401

lib()  # fail at import time when the HIP extension is missing: there is no CPU fallback

from .linsys import BatchLinsys, CscPattern, HipLDLSolver, symbolic_analyze  # noqa: E402,F401
from .osqp_batch import OSQPBatch, OSQPHorizon, STATUS_NAMES, default_settings  # noqa: E402,F401
from .groups import OSQPBatchGroups  # noqa: E402,F401
from . import workloads  # noqa: E402,F401

__all__ = ["BatchLinsys", "CscPattern", "HipLDLSolver", "OSQPBatch", "OSQPBatchGroups", "OSQPHorizon", "STATUS_NAMES", "default_settings",
           "symbolic_analyze", "workloads", "build", "lib"]

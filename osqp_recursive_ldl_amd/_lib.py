"""Loader for libosqp_rldl_hip.so (the C-ABI of include/osqp_rldl_hip.h) via ctypes.

There is no CPU fallback anywhere in this package: if the shared library is missing the import of
this module raises, and if no HIP device is present every compute entry point returns
RLDL_NO_DEVICE_ERROR (100), which the wrappers turn into a RuntimeError.
"""
import ctypes as C
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libosqp_rldl_hip.so")

c_int = C.c_longlong
c_float = C.c_double
IP = C.POINTER(c_int)
FP = C.POINTER(c_float)
VP = C.c_void_p

RLDL_NO_DEVICE_ERROR = 100
HIP_LDL_SOLVER = 20


class Csc(C.Structure):
    """include/types.h:21-29"""
    _fields_ = [("nzmax", c_int), ("m", c_int), ("n", c_int), ("p", IP), ("i", IP), ("x", FP), ("nz", c_int)]


class HipldlSolver(C.Structure):
    """Prefix-compatible with struct linsys_solver (include/types.h:298-319)."""
    pass


HipldlSolver._fields_ = [
    ("type", C.c_int),
    ("solve", C.CFUNCTYPE(c_int, C.POINTER(HipldlSolver), FP)),
    ("free", C.CFUNCTYPE(None, C.POINTER(HipldlSolver))),
    ("update_matrices", C.CFUNCTYPE(c_int, C.POINTER(HipldlSolver), C.POINTER(Csc), C.POINTER(Csc))),
    ("update_rho_vec", C.CFUNCTYPE(c_int, C.POINTER(HipldlSolver), FP)),
    ("nthreads", c_int),
    ("impl", VP),
]


class OSQPBatchSettings(C.Structure):
    _fields_ = [("rho", c_float), ("sigma", c_float), ("alpha", c_float), ("eps_abs", c_float),
                ("eps_rel", c_float), ("eps_prim_inf", c_float), ("eps_dual_inf", c_float),
                ("max_iter", c_int), ("check_termination", c_int), ("warm_start", c_int),
                ("scaling", c_int), ("scaled_termination", c_int), ("adaptive_rho", c_int),
                ("adaptive_rho_interval", c_int), ("adaptive_rho_tolerance", c_float),
                ("polish", c_int), ("polish_refine_iter", c_int), ("delta", c_float)]


class StageDims(C.Structure):
    _fields_ = [("N", c_int), ("nx", c_int), ("nu", c_int), ("ny", c_int), ("nt", c_int)]


# every symbol include/osqp_rldl_hip.h declares
EXPORTED = [
    "init_linsys_solver_hipldl", "solve_linsys_hipldl", "update_linsys_solver_matrices_hipldl",
    "update_linsys_solver_rho_vec_hipldl", "free_linsys_solver_hipldl",
    "rldl_batch_init", "rldl_batch_solve", "rldl_batch_update_matrices", "rldl_batch_update_rho_vec",
    "rldl_batch_free", "rldl_batch_dims", "rldl_batch_export_symbolic", "rldl_batch_export_factor",
    "rldl_batch_factor_status", "rldl_batch_export_prod", "rldl_batch_time_solve", "rldl_batch_trace_solve", "rldl_batch_trace_factor", "rldl_batch_set_cache_policy", "rldl_batch_time_solve_rotating",
    "osqp_batch_set_default_settings", "osqp_batch_setup", "osqp_batch_solve", "osqp_batch_update_lin_cost",
    "osqp_groups_bucket", "osqp_batch_pack_results", "osqp_batch_update_bounds", "osqp_batch_update_bounds_async", "osqp_batch_partial_update_bounds_async", "osqp_batch_update_rho", "osqp_batch_update_settings", "osqp_batch_update_P_A", "osqp_batch_update_P_A_async", "osqp_batch_warm_start",
    "osqp_batch_get", "osqp_batch_get_iterates", "osqp_batch_get_scaling", "osqp_batch_get_polish_status", "osqp_batch_get_rho", "osqp_batch_linsys", "osqp_batch_solve_async", "osqp_batch_wait", "osqp_batch_setup_recursive", "osqp_batch_update_recursive", "osqp_batch_partial_update_bounds",
    "osqp_batch_time_iteration", "osqp_batch_last_loop", "osqp_batch_trace_iteration", "osqp_batch_cleanup",
    "rldl_batch_init_recursive", "rldl_batch_update_from_stage", "rldl_version",
    "rldl_symbolic_analyze", "rldl_stage_permutation", "rldl_plan_export", "rldl_stage_prod_export", "rldl_setup_AP_matrices", "rldl_csc_free",
    "osqp_horizon_setup", "osqp_horizon_update", "osqp_horizon_workspace", "osqp_horizon_N", "osqp_horizon_last_update",
    "osqp_horizon_free", "osqp_batch_multi_key", "osqp_dist_unique_id", "osqp_dist_init", "osqp_dist_record_len", "osqp_dist_gather_results", "osqp_dist_free", "osqp_horizon_is_single", "osqp_horizon_workspaces", "osqp_horizon_ld", "osqp_horizon_update_P_A", "osqp_horizon_warm_start",
    "osqp_multi_create", "osqp_multi_solve", "osqp_multi_update_P_A", "osqp_multi_get", "osqp_multi_free",
]


def build(force=False):
    """Compile the host C + gfx950 HIP sources in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".c", ".h", ".hip"))]
    srcs.append(os.path.join(PKG_DIR, "..", "include", "osqp_rldl_hip.h"))
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", CSRC])
    return LIB_PATH


def _declare(L):
    PC = C.POINTER(Csc)
    L.init_linsys_solver_hipldl.argtypes = [C.POINTER(C.POINTER(HipldlSolver)), PC, PC, c_float, FP, c_int]
    L.init_linsys_solver_hipldl.restype = c_int
    L.solve_linsys_hipldl.argtypes = [C.POINTER(HipldlSolver), FP]
    L.solve_linsys_hipldl.restype = c_int
    L.update_linsys_solver_matrices_hipldl.argtypes = [C.POINTER(HipldlSolver), PC, PC]
    L.update_linsys_solver_matrices_hipldl.restype = c_int
    L.update_linsys_solver_rho_vec_hipldl.argtypes = [C.POINTER(HipldlSolver), FP]
    L.update_linsys_solver_rho_vec_hipldl.restype = c_int
    L.free_linsys_solver_hipldl.argtypes = [C.POINTER(HipldlSolver)]
    L.free_linsys_solver_hipldl.restype = None

    L.rldl_batch_init.argtypes = [C.POINTER(VP), c_int, PC, PC, VP, VP, c_float, VP, c_int, IP, VP]
    L.rldl_batch_init.restype = c_int
    L.rldl_batch_solve.argtypes = [VP, VP]
    L.rldl_batch_solve.restype = c_int
    L.rldl_batch_update_matrices.argtypes = [VP, VP, VP]
    L.rldl_batch_update_matrices.restype = c_int
    L.rldl_batch_update_rho_vec.argtypes = [VP, VP, VP]
    L.rldl_batch_update_rho_vec.restype = c_int
    L.rldl_batch_free.argtypes = [VP]
    L.rldl_batch_free.restype = None
    L.rldl_batch_dims.argtypes = [VP, IP, IP, IP, IP, IP]
    L.rldl_batch_dims.restype = c_int
    L.rldl_batch_export_symbolic.argtypes = [VP] + [IP] * 10
    L.rldl_batch_export_symbolic.restype = c_int
    L.rldl_batch_export_factor.argtypes = [VP, c_int, FP, FP, FP, FP]
    L.rldl_batch_export_factor.restype = c_int
    L.rldl_batch_export_prod.argtypes = [VP, c_int, IP, VP, VP, VP, VP, VP, FP]
    L.rldl_batch_export_prod.restype = c_int
    L.rldl_stage_prod_export.argtypes = [VP, VP, C.POINTER(StageDims), IP, VP, VP, VP, VP, VP, IP]
    L.rldl_stage_prod_export.restype = c_int
    L.osqp_multi_create.argtypes = [C.POINTER(VP), C.POINTER(VP), c_int, IP, VP]
    L.osqp_multi_create.restype = c_int
    L.osqp_multi_update_P_A.argtypes = [VP, C.POINTER(VP), C.POINTER(VP)]
    L.osqp_multi_update_P_A.restype = c_int
    L.osqp_multi_solve.argtypes = [VP]
    L.osqp_multi_solve.restype = c_int
    L.osqp_multi_get.argtypes = [VP] + [VP] * 8
    L.osqp_multi_get.restype = c_int
    L.osqp_multi_free.argtypes = [VP]
    L.osqp_multi_free.restype = None
    L.rldl_batch_factor_status.argtypes = [VP, IP]
    L.rldl_batch_factor_status.restype = c_int
    L.rldl_batch_time_solve.argtypes = [VP, VP, c_int, FP]
    L.rldl_batch_time_solve.restype = c_int
    L.rldl_batch_time_solve_rotating.argtypes = [VP, VP, c_int, c_int, FP]
    L.rldl_batch_time_solve_rotating.restype = c_int
    L.rldl_batch_trace_solve.argtypes = [VP, VP, VP]
    L.rldl_batch_trace_solve.restype = c_int
    L.rldl_batch_set_cache_policy.argtypes = [VP, c_int]
    L.rldl_batch_set_cache_policy.restype = c_int
    L.rldl_batch_trace_factor.argtypes = [VP, VP]
    L.rldl_batch_trace_factor.restype = c_int

    L.osqp_batch_set_default_settings.argtypes = [C.POINTER(OSQPBatchSettings)]
    L.osqp_batch_set_default_settings.restype = None
    L.osqp_batch_setup.argtypes = [C.POINTER(VP), c_int, PC, PC, VP, VP, VP, VP, VP, C.POINTER(OSQPBatchSettings), IP, VP]
    L.osqp_batch_setup.restype = c_int
    L.osqp_batch_solve.argtypes = [VP]
    L.osqp_batch_solve.restype = c_int
    L.osqp_batch_update_lin_cost.argtypes = [VP, VP]
    L.osqp_batch_update_lin_cost.restype = c_int
    L.osqp_batch_pack_results.argtypes = [VP, VP]
    L.osqp_batch_pack_results.restype = c_int
    L.osqp_groups_bucket.argtypes = [c_int, VP, VP, IP]
    L.osqp_groups_bucket.restype = c_int
    L.osqp_batch_update_bounds.argtypes = [VP, VP, VP]
    L.osqp_batch_update_bounds.restype = c_int
    L.osqp_batch_update_bounds_async.argtypes = [VP, VP, VP]
    L.osqp_batch_update_bounds_async.restype = c_int
    L.osqp_batch_partial_update_bounds_async.argtypes = [VP, c_int, c_int, VP, VP]
    L.osqp_batch_partial_update_bounds_async.restype = c_int
    L.osqp_batch_update_rho.argtypes = [VP, c_float]
    L.osqp_batch_update_rho.restype = c_int
    L.osqp_batch_update_P_A.argtypes = [VP, VP, VP]
    L.osqp_batch_update_P_A.restype = c_int
    L.osqp_batch_update_P_A_async.argtypes = [VP, VP, VP]
    L.osqp_batch_update_P_A_async.restype = c_int
    L.osqp_batch_warm_start.argtypes = [VP, VP, VP]
    L.osqp_batch_warm_start.restype = c_int
    L.osqp_batch_get.argtypes = [VP] + [C.POINTER(VP)] * 8
    L.osqp_batch_get.restype = c_int
    L.osqp_batch_get_iterates.argtypes = [VP] + [C.POINTER(VP)] * 5
    L.osqp_batch_get_iterates.restype = c_int
    L.osqp_batch_get_scaling.argtypes = [VP] + [C.POINTER(VP)] * 3
    L.osqp_batch_get_scaling.restype = c_int
    L.osqp_batch_update_settings.argtypes = [VP, C.POINTER(OSQPBatchSettings)]
    L.osqp_batch_update_settings.restype = c_int
    L.osqp_batch_get_polish_status.argtypes = [VP, C.POINTER(VP)]
    L.osqp_batch_get_polish_status.restype = c_int
    L.osqp_batch_linsys.argtypes = [VP]
    L.osqp_batch_linsys.restype = VP
    L.osqp_batch_time_iteration.argtypes = [VP, c_int, FP]
    L.osqp_batch_time_iteration.restype = c_int
    L.osqp_batch_solve_async.argtypes = [VP]
    L.osqp_batch_solve_async.restype = c_int
    L.osqp_batch_wait.argtypes = [VP]
    L.osqp_batch_wait.restype = c_int
    L.osqp_batch_setup_recursive.argtypes = [C.POINTER(VP), c_int, C.POINTER(StageDims)] + [C.POINTER(Csc)] * 7 + [VP, VP, VP,
                                             C.POINTER(OSQPBatchSettings), C.POINTER(C.POINTER(Csc)), C.POINTER(C.POINTER(Csc)), VP]
    L.osqp_batch_setup_recursive.restype = c_int
    L.osqp_batch_update_recursive.argtypes = [VP, c_int, VP, VP]
    L.osqp_batch_update_recursive.restype = c_int
    L.osqp_batch_partial_update_bounds.argtypes = [VP, c_int, c_int, VP, VP]
    L.osqp_batch_partial_update_bounds.restype = c_int
    L.osqp_batch_last_loop.argtypes = [VP, FP, IP, IP]
    L.osqp_batch_last_loop.restype = c_int
    L.osqp_batch_trace_iteration.argtypes = [VP, c_int, VP]
    L.osqp_batch_trace_iteration.restype = c_int
    L.osqp_batch_cleanup.argtypes = [VP]
    L.osqp_batch_cleanup.restype = None

    L.rldl_batch_init_recursive.argtypes = [C.POINTER(VP), c_int, C.POINTER(StageDims), PC, PC, VP, VP, c_float, VP, VP]
    L.rldl_batch_init_recursive.restype = c_int
    L.rldl_batch_update_from_stage.argtypes = [VP, c_int, VP, VP, VP]
    L.rldl_batch_update_from_stage.restype = c_int
    L.rldl_version.argtypes = []
    L.rldl_version.restype = C.c_char_p
    L.rldl_stage_permutation.argtypes = [c_int] * 5 + [IP]
    L.rldl_stage_permutation.restype = None
    L.rldl_symbolic_analyze.argtypes = [PC, PC, c_int, IP, IP, IP, IP] + [IP] * 10
    L.rldl_symbolic_analyze.restype = c_int
    L.rldl_plan_export.argtypes = [PC, PC, c_int, IP, IP, C.POINTER(C.c_int), c_int, IP]
    L.rldl_plan_export.restype = c_int
    L.rldl_setup_AP_matrices.argtypes = [C.POINTER(StageDims)] + [PC] * 7 + [C.POINTER(PC), C.POINTER(PC)] + [IP] * 6
    L.rldl_setup_AP_matrices.restype = c_int
    L.rldl_csc_free.argtypes = [PC]
    L.rldl_csc_free.restype = None
    L.rldl_device_available.restype = C.c_int
    L.osqp_batch_get_rho.argtypes = [VP, C.POINTER(VP), C.POINTER(VP), C.POINTER(VP)]
    L.osqp_batch_get_rho.restype = c_int
    L.osqp_horizon_setup.argtypes = [C.POINTER(VP), c_int, C.POINTER(StageDims), c_int] + [PC] * 7 + [VP, VP, VP,
                                     C.POINTER(OSQPBatchSettings), VP]
    L.osqp_horizon_setup.restype = c_int
    L.osqp_horizon_update.argtypes = [VP, c_int, VP, VP, VP]
    L.osqp_horizon_update.restype = c_int
    L.osqp_horizon_workspace.argtypes = [VP]
    L.osqp_horizon_workspace.restype = VP
    L.osqp_horizon_N.argtypes = [VP]
    L.osqp_horizon_N.restype = c_int
    L.osqp_batch_multi_key.argtypes = [VP]
    L.osqp_batch_multi_key.restype = c_int
    L.osqp_dist_unique_id.argtypes = [C.c_char_p]
    L.osqp_dist_unique_id.restype = c_int
    L.osqp_dist_init.argtypes = [C.POINTER(VP), C.c_char_p, c_int, c_int, VP]
    L.osqp_dist_init.restype = c_int
    L.osqp_dist_record_len.argtypes = [VP]
    L.osqp_dist_record_len.restype = c_int
    L.osqp_dist_gather_results.argtypes = [VP, VP, VP]
    L.osqp_dist_gather_results.restype = c_int
    L.osqp_dist_free.argtypes = [VP]
    L.osqp_dist_free.restype = None
    for name in ("osqp_horizon_is_single", "osqp_horizon_workspaces"):
        getattr(L, name).argtypes = [VP]
        getattr(L, name).restype = c_int
    L.osqp_horizon_ld.argtypes = [VP, IP, IP]
    L.osqp_horizon_ld.restype = c_int
    L.osqp_horizon_update_P_A.argtypes = [VP, VP, VP]
    L.osqp_horizon_update_P_A.restype = c_int
    L.osqp_horizon_warm_start.argtypes = [VP, VP, VP]
    L.osqp_horizon_warm_start.restype = c_int
    L.osqp_horizon_last_update.argtypes = [VP, IP, IP, IP]
    L.osqp_horizon_last_update.restype = c_int
    L.osqp_horizon_free.argtypes = [VP]
    L.osqp_horizon_free.restype = None


_lib = None


def lib():
    """The loaded C-ABI library.  Raises (never falls back) when the HIP extension is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libosqp_rldl_hip.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C osqp_recursive_ldl_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
        try:
            # PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  Two HIP runtimes in one
            # process cannot both open the device, so torch's copy must be the one already loaded when our
            # library's DT_NEEDED libamdhip64.so.7 is resolved (it is then reused by soname).
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        missing = [s for s in EXPORTED if not hasattr(L, s)]
        if missing:
            raise RuntimeError("libosqp_rldl_hip.so lacks symbols: %s" % missing)
        _declare(L)
        _lib = L
    return _lib


def check(rc, what):
    if rc == RLDL_NO_DEVICE_ERROR:
        raise RuntimeError("%s: no HIP device available (the backend has no CPU fallback)" % what)
    return rc

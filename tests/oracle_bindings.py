"""ctypes bindings for the CPU oracle (oracle/liboracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
LIB_PATH_OFAST = os.path.join(ORACLE_DIR, "liboracle_ofast.so")   # same sources, the reference's shipped flags (bench.py only)

c_int = C.c_longlong
c_float = C.c_double
IP = C.POINTER(c_int)
FP = C.POINTER(c_float)


class Csc(C.Structure):
    _fields_ = [("nzmax", c_int), ("m", c_int), ("n", c_int), ("p", IP), ("i", IP), ("x", FP), ("nz", c_int)]


class Settings(C.Structure):
    _fields_ = [("rho", c_float), ("sigma", c_float), ("alpha", c_float), ("eps_abs", c_float),
                ("eps_rel", c_float), ("eps_prim_inf", c_float), ("eps_dual_inf", c_float),
                ("max_iter", c_int), ("check_termination", c_int), ("warm_start", c_int),
                ("scaling", c_int), ("scaled_termination", c_int), ("adaptive_rho", c_int),
                ("adaptive_rho_interval", c_int), ("adaptive_rho_tolerance", c_float),
                ("polish", c_int), ("polish_refine_iter", c_int), ("delta", c_float)]


class StageDims(C.Structure):
    _fields_ = [("N", c_int), ("nx", c_int), ("nu", c_int), ("ny", c_int), ("nt", c_int)]


class Info(C.Structure):
    _fields_ = [("iter", c_int), ("status_val", c_int), ("rho_updates", c_int), ("obj_val", c_float),
                ("pri_res", c_float), ("dua_res", c_float), ("rho_estimate", c_float), ("status_polish", c_int)]


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or not os.path.exists(LIB_PATH_OFAST) or any(
            os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(LIB_PATH)
            for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        VP = C.c_void_p
        L.orc_linsys_init.argtypes = [C.POINTER(VP), C.POINTER(Csc), C.POINTER(Csc), c_float, FP, c_int, IP]
        L.orc_linsys_init.restype = c_int
        L.orc_linsys_solve.argtypes = [VP, FP]
        L.orc_linsys_solve.restype = c_int
        L.orc_linsys_update_matrices.argtypes = [VP, C.POINTER(Csc), C.POINTER(Csc)]
        L.orc_linsys_update_matrices.restype = c_int
        L.orc_linsys_update_rho_vec.argtypes = [VP, FP]
        L.orc_linsys_update_rho_vec.restype = c_int
        L.orc_linsys_free.argtypes = [VP]
        L.orc_linsys_nnzL.argtypes = [VP]
        L.orc_linsys_nnzL.restype = c_int
        L.orc_linsys_nnzKKT.argtypes = [VP]
        L.orc_linsys_nnzKKT.restype = c_int
        L.orc_linsys_export.argtypes = [VP, IP, IP, IP, IP, IP, FP, FP, FP]
        L.orc_linsys_export_KKT.argtypes = [VP, IP, IP, FP]
        L.orc_form_KKT.argtypes = [C.POINTER(Csc), C.POINTER(Csc), c_float, FP, IP, IP, C.POINTER(IP), IP, IP]
        L.orc_form_KKT.restype = C.POINTER(Csc)
        L.orc_update_KKT_P.argtypes = [C.POINTER(Csc), C.POINTER(Csc), IP, c_float, IP, c_int]
        L.orc_update_KKT_A.argtypes = [C.POINTER(Csc), C.POINTER(Csc), IP]
        L.orc_csc_spfree.argtypes = [C.POINTER(Csc)]
        L.orc_qdldl_etree.argtypes = [c_int, IP, IP, IP, IP, IP]
        L.orc_qdldl_etree.restype = c_int
        L.orc_min_degree_order.argtypes = [c_int, IP, IP, IP]
        L.orc_set_default_settings.argtypes = [C.POINTER(Settings)]
        L.orc_setup.argtypes = [C.POINTER(VP), C.POINTER(Csc), FP, C.POINTER(Csc), FP, FP, C.POINTER(Settings), IP]
        L.orc_setup.restype = c_int
        L.orc_solve.argtypes = [VP]
        L.orc_solve.restype = c_int
        L.orc_cleanup.argtypes = [VP]
        L.orc_update_lin_cost.argtypes = [VP, FP]
        L.orc_update_bounds.argtypes = [VP, FP, FP]
        L.orc_update_bounds.restype = c_int
        L.orc_update_rho.argtypes = [VP, c_float]
        L.orc_update_rho.restype = c_int
        L.orc_update_P_A.argtypes = [VP, FP, FP]
        L.orc_update_P_A.restype = c_int
        L.orc_warm_start.argtypes = [VP, FP, FP]
        for nm in ("orc_ws_x", "orc_ws_y", "orc_ws_z", "orc_ws_sol_x", "orc_ws_sol_y", "orc_ws_delta_x", "orc_ws_delta_y",
                   "orc_ws_D", "orc_ws_E"):
            getattr(L, nm).argtypes = [VP]
            getattr(L, nm).restype = FP
        L.orc_ws_info.argtypes = [VP]
        L.orc_ws_info.restype = C.POINTER(Info)
        L.orc_ws_linsys.argtypes = [VP]
        L.orc_ws_linsys.restype = VP
        L.orc_use_external_linsys.argtypes = [VP, VP, VP, VP]
        L.orc_use_external_linsys.restype = None
        L.orc_ws_P.argtypes = [VP]; L.orc_ws_P.restype = C.POINTER(Csc)
        L.orc_ws_A.argtypes = [VP]; L.orc_ws_A.restype = C.POINTER(Csc)
        L.orc_ws_rho_vec.argtypes = [VP]; L.orc_ws_rho_vec.restype = FP
        L.orc_bench_shared_pattern.argtypes = [c_int, c_int, c_int, IP, IP, FP, IP, IP, FP, FP, FP, FP,
                                               C.POINTER(Settings), IP, FP, FP, C.POINTER(C.c_double),
                                               C.POINTER(C.c_double)]
        L.orc_bench_shared_pattern.restype = C.c_double
        _bind_bench_mt(L)
        L.orc_rldl_xeven_stride.argtypes = [C.POINTER(StageDims)]
        L.orc_rldl_xeven_stride.restype = c_int
        L.orc_rldl_factor.argtypes = [C.POINTER(StageDims), C.POINTER(Csc), C.POINTER(Csc), c_float, FP, c_int, c_int, c_int, c_int, FP,
                                      IP, IP, FP, c_int, FP, IP]
        L.orc_rldl_factor.restype = c_int
        L.orc_rldl_border.argtypes = [c_int, IP, IP, FP, FP, c_int, FP, FP, FP, FP]
        L.orc_rldl_border.restype = None
        _lib = L
    return _lib


def _bind_bench_mt(L):
    L.orc_bench_shared_pattern_mt.argtypes = [c_int, c_int, c_int, c_int, c_int, IP, IP, FP, IP, IP, FP, FP, FP, FP,
                                              C.POINTER(Settings), IP]
    L.orc_bench_shared_pattern_mt.restype = C.c_double


_lib_ofast = None


def lib_ofast():
    """The -Ofast build (bench.py's cpu_baseline): only the two bench entry points are bound."""
    global _lib_ofast
    if _lib_ofast is None:
        build()
        L = C.CDLL(LIB_PATH_OFAST)
        L.orc_bench_shared_pattern.argtypes = [c_int, c_int, c_int, IP, IP, FP, IP, IP, FP, FP, FP, FP,
                                               C.POINTER(Settings), IP, FP, FP, C.POINTER(C.c_double),
                                               C.POINTER(C.c_double)]
        L.orc_bench_shared_pattern.restype = C.c_double
        _bind_bench_mt(L)
        _lib_ofast = L
    return _lib_ofast


def ip(a):
    return a.ctypes.data_as(IP) if a is not None else None


def fp(a):
    return a.ctypes.data_as(FP) if a is not None else None


class CscHolder:
    """Keeps numpy arrays alive behind an orc_csc struct."""

    def __init__(self, m, n, p, i, x):
        self.p = np.ascontiguousarray(p, dtype=np.int64)
        self.i = np.ascontiguousarray(i, dtype=np.int64)
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        self.m, self.n = int(m), int(n)
        self.s = Csc(len(self.x), self.m, self.n, ip(self.p), ip(self.i), fp(self.x), -1)

    @classmethod
    def from_scipy(cls, M):
        from scipy import sparse
        M = sparse.csc_matrix(M)
        M.sort_indices()
        return cls(M.shape[0], M.shape[1], M.indptr, M.indices, M.data)

    @property
    def ref(self):
        return C.byref(self.s)


def settings(**kw):
    s = Settings()
    lib().orc_set_default_settings(C.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise KeyError(k)
        setattr(s, k, v)
    return s


class OracleLinsys:
    """init / solve / update_rho_vec / update_matrices of the reference's qdldl backend (CPU oracle)."""

    def __init__(self, P, A, sigma, rho_vec, polish=0, perm=None):
        self.Pc, self.Ac = CscHolder.from_scipy(P), CscHolder.from_scipy(A)
        self.n, self.m = self.Pc.n, self.Ac.m
        self.h = C.c_void_p()
        rv = None if rho_vec is None else np.ascontiguousarray(rho_vec, dtype=np.float64)
        pm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int64)
        self.status = lib().orc_linsys_init(C.byref(self.h), self.Pc.ref, self.Ac.ref, sigma, fp(rv), polish, ip(pm))

    def solve(self, b):
        b = np.array(b, dtype=np.float64, copy=True)
        lib().orc_linsys_solve(self.h, fp(b))
        return b

    def update_rho_vec(self, rho_vec):
        rv = np.ascontiguousarray(rho_vec, dtype=np.float64)
        return lib().orc_linsys_update_rho_vec(self.h, fp(rv))

    def update_matrices(self, P, A):
        Pc, Ac = CscHolder.from_scipy(P), CscHolder.from_scipy(A)
        return lib().orc_linsys_update_matrices(self.h, Pc.ref, Ac.ref)

    def export(self):
        N = self.n + self.m
        nz = lib().orc_linsys_nnzL(self.h)
        P = np.zeros(N, np.int64); et = np.zeros(N, np.int64); Lnz = np.zeros(N, np.int64)
        Lp = np.zeros(N + 1, np.int64); Li = np.zeros(max(nz, 1), np.int64); Lx = np.zeros(max(nz, 1))
        D = np.zeros(N); Dinv = np.zeros(N)
        lib().orc_linsys_export(self.h, ip(P), ip(et), ip(Lnz), ip(Lp), ip(Li), fp(Lx), fp(D), fp(Dinv))
        return dict(P=P, etree=et, Lnz=Lnz, Lp=Lp, Li=Li[:nz], Lx=Lx[:nz], D=D, Dinv=Dinv)

    def export_KKT(self):
        N = self.n + self.m
        nz = lib().orc_linsys_nnzKKT(self.h)
        Kp = np.zeros(N + 1, np.int64); Ki = np.zeros(nz, np.int64); Kx = np.zeros(nz)
        lib().orc_linsys_export_KKT(self.h, ip(Kp), ip(Ki), fp(Kx))
        return Kp, Ki, Kx

    def free(self):
        if self.h:
            lib().orc_linsys_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class OracleRLDL:
    """The stage recursion of src/recursive_ldl.c (oracle/rldl_oracle.c): LDL_factorize_recursive on the assembled P, A of an
    MPC problem, LDL_update_from_pivot restarts from the cached constraint block of a stage."""

    def __init__(self, dims, mirror_drops=0, terminal_rho_own=1, Nmax=None):
        self.d = StageDims(*[int(v) for v in dims])
        N, nx, nu, ny, nt = [int(v) for v in dims]
        self.n, self.m = N * (nx + nu), N * (nx + ny) + nt
        self.mirror, self.own, self.Nmax = int(mirror_drops), int(terminal_rho_own), int(N if Nmax is None else Nmax)
        Nk = self.n + self.m
        smax = max(nx + ny, nx + nu, nt)
        self.cap = (2 * N + 2) * 2 * smax * smax
        self.xeven = np.zeros((N + 1) * int(lib().orc_rldl_xeven_stride(C.byref(self.d))))
        self.Lp = np.zeros(Nk + 1, np.int64); self.Li = np.zeros(self.cap, np.int64); self.Lx = np.zeros(self.cap)
        self.Dinv = np.zeros(Nk); self.perm = np.zeros(Nk, np.int64)
        self.nnz = -100

    def factor(self, P, A, sigma, rho_inv, iter_start=-1):
        Pc, Ac = CscHolder.from_scipy(P), CscHolder.from_scipy(A)
        ri = np.ascontiguousarray(rho_inv, dtype=np.float64)
        self.nnz = int(lib().orc_rldl_factor(C.byref(self.d), Pc.ref, Ac.ref, sigma, fp(ri), self.Nmax, self.mirror, self.own,
                                             int(iter_start), fp(self.xeven), ip(self.Lp), ip(self.Li), fp(self.Lx), self.cap,
                                             fp(self.Dinv), ip(self.perm)))
        return self.nnz

    def L(self):
        from scipy import sparse
        Nk = self.n + self.m
        return sparse.csc_matrix((self.Lx[:self.nnz].copy(), self.Li[:self.nnz].copy(), self.Lp.copy()), shape=(Nk, Nk))


def rldl_border(Lp, Li, Lx, Dinv, V, Y):
    """compute_Vhat's border algebra (oracle/rldl_oracle.c: orc_rldl_border): (V^ = V L^-T D^-1, Y^ = Y - V^ D V^') for the rows V
    (dense [nrows, nf]) of a block coupled to a factorised part (strictly lower CSC L, Dinv, nf x nf)."""
    nf, nrows = len(Dinv), V.shape[0]
    Lp = np.ascontiguousarray(Lp, np.int64); Li = np.ascontiguousarray(Li, np.int64); Lx = np.ascontiguousarray(Lx, np.float64)
    Dv = np.ascontiguousarray(Dinv, np.float64)
    Vc = np.ascontiguousarray(V, np.float64); Yc = np.ascontiguousarray(Y, np.float64)
    Vh, Yh = np.zeros_like(Vc), np.zeros_like(Yc)
    lib().orc_rldl_border(nf, ip(Lp), ip(Li), fp(Lx), fp(Dv), nrows, fp(Vc), fp(Yc), fp(Vh), fp(Yh))
    return Vh, Yh


class OracleOSQP:
    """osqp_setup / osqp_solve / osqp_update_* on the CPU oracle."""

    def __init__(self, P, q, A, l, u, perm=None, **kw):
        self.Pc, self.Ac = CscHolder.from_scipy(P), CscHolder.from_scipy(A)
        self.n, self.m = self.Pc.n, self.Ac.m
        self.q = np.ascontiguousarray(q, dtype=np.float64)
        self.l = np.ascontiguousarray(l, dtype=np.float64)
        self.u = np.ascontiguousarray(u, dtype=np.float64)
        self.st = settings(**kw)
        pm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int64)
        self.h = C.c_void_p()
        self.status = lib().orc_setup(C.byref(self.h), self.Pc.ref, fp(self.q), self.Ac.ref, fp(self.l),
                                      fp(self.u), C.byref(self.st), ip(pm))

    def _vec(self, fn, n):
        return np.ctypeslib.as_array(fn(self.h), shape=(max(n, 1),))[:n].copy()

    def solve(self):
        flag = lib().orc_solve(self.h)
        info = lib().orc_ws_info(self.h).contents
        return dict(flag=flag, x=self._vec(lib().orc_ws_sol_x, self.n), y=self._vec(lib().orc_ws_sol_y, self.m),
                    iter=info.iter, status=info.status_val, obj=info.obj_val, pri_res=info.pri_res,
                    dua_res=info.dua_res, rho_updates=info.rho_updates, status_polish=info.status_polish,
                    x_iter=self._vec(lib().orc_ws_x, self.n), y_iter=self._vec(lib().orc_ws_y, self.m),
                    z_iter=self._vec(lib().orc_ws_z, self.m), delta_x=self._vec(lib().orc_ws_delta_x, self.n),
                    delta_y=self._vec(lib().orc_ws_delta_y, self.m))

    def backend_data(self):
        """What the workspace hands to init_linsys_solver (osqp.c:157-160): the (scaled) P, A, sigma and rho_vec."""
        from scipy import sparse

        def mat(cp):
            c = cp.contents
            nz = int(c.p[c.n])
            return sparse.csc_matrix((np.array(c.x[:nz], float), np.array(c.i[:nz], np.int64), np.array(c.p[:c.n + 1], np.int64)),
                                     shape=(int(c.m), int(c.n)))
        rv = np.array(lib().orc_ws_rho_vec(self.h)[:self.m], float)
        return mat(lib().orc_ws_P(self.h)), mat(lib().orc_ws_A(self.h)), float(self.st.sigma), rv

    def use_external_linsys(self, self_ptr, solve_fn, update_rho_vec_fn):
        """Route the ADMM loop's linear solves through an external plugin object with the reference's vtable shape
        (work->linsys_solver->solve / ->update_rho_vec, auxil.c:185, osqp.c:1310-1318)."""
        cast = lambda f: C.cast(f, C.c_void_p)
        lib().orc_use_external_linsys(self.h, C.cast(self_ptr, C.c_void_p), cast(solve_fn), cast(update_rho_vec_fn))

    def scaling_vectors(self):
        lib().orc_ws_c.restype = C.c_double
        lib().orc_ws_c.argtypes = [C.c_void_p]
        return self._vec(lib().orc_ws_D, self.n), self._vec(lib().orc_ws_E, self.m), float(lib().orc_ws_c(self.h))

    def update_lin_cost(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64)
        lib().orc_update_lin_cost(self.h, fp(q))

    def update_bounds(self, l, u):
        l = np.ascontiguousarray(l, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
        return lib().orc_update_bounds(self.h, fp(l), fp(u))

    def update_rho(self, rho):
        return lib().orc_update_rho(self.h, rho)

    def update_P_A(self, Px=None, Ax=None):
        Px = None if Px is None else np.ascontiguousarray(Px, dtype=np.float64)
        Ax = None if Ax is None else np.ascontiguousarray(Ax, dtype=np.float64)
        return lib().orc_update_P_A(self.h, fp(Px), fp(Ax))

    def warm_start(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        lib().orc_warm_start(self.h, fp(x), fp(y))

    def linsys_export(self):
        ls = lib().orc_ws_linsys(self.h)
        tmp = OracleLinsys.__new__(OracleLinsys)
        tmp.h = C.c_void_p(ls); tmp.n, tmp.m = self.n, self.m
        out = tmp.export()
        tmp.h = C.c_void_p()
        return out

    def cleanup(self):
        if self.h:
            lib().orc_cleanup(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass

"""Batched, device-resident mirror of the reference's public solver API for the hot path:
osqp_setup / osqp_solve / osqp_update_lin_cost / osqp_update_bounds / osqp_update_rho /
osqp_update_P_A / osqp_warm_start / osqp_cleanup (src/osqp.c), one instance per wavefront.
"""
import ctypes as C

import numpy as np

from . import _lib
from .linsys import BatchLinsys, CscPattern, _dev_f64, _dptr, _ip

STATUS_NAMES = {1: "solved", 2: "solved inaccurate", 3: "primal infeasible inaccurate", 4: "dual infeasible inaccurate",
                -2: "maximum iterations reached", -3: "primal infeasible", -4: "dual infeasible", -7: "problem non convex",
                -10: "unsolved"}


def default_settings(**kw):
    s = _lib.OSQPBatchSettings()
    _lib.lib().osqp_batch_set_default_settings(C.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise KeyError("unknown setting %r" % k)
        setattr(s, k, v)
    return s


class OSQPBatch:
    def __init__(self, P_pattern, A_pattern, Px, Ax, q, l, u, perm=None, stream=None, **settings):
        """stream: optional torch.cuda.Stream the workspace enqueues on (default: the legacy default stream)."""
        L = _lib.lib()
        self._stream = stream
        self.P = P_pattern if isinstance(P_pattern, CscPattern) else CscPattern(P_pattern)
        self.A = A_pattern if isinstance(A_pattern, CscPattern) else CscPattern(A_pattern)
        self.n, self.m = self.P.shape[0], self.A.shape[0]
        self.batch = int(q.shape[0])
        _dev_f64(Px, (self.batch, self.P.nnz), "Px"); _dev_f64(Ax, (self.batch, self.A.nnz), "Ax")
        _dev_f64(q, (self.batch, self.n), "q"); _dev_f64(l, (self.batch, self.m), "l"); _dev_f64(u, (self.batch, self.m), "u")
        self.settings = default_settings(**settings)
        pm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int64)
        self.h = C.c_void_p()
        self.status = _lib.check(L.osqp_batch_setup(C.byref(self.h), self.batch, self.P.ref, self.A.ref, _dptr(Px), _dptr(Ax),
                                                    _dptr(q), _dptr(l), _dptr(u), C.byref(self.settings),
                                                    None if pm is None else _ip(pm),
                                                    None if stream is None else C.c_void_p(stream.cuda_stream)), "osqp_batch_setup")
        self._device = q.device

    @classmethod
    def recursive(cls, dims, Q0, Qi, QN, A0, Ai, Aij, AN, q, l, u, **settings):
        """osqp_setup_recursive (src/recursive_ldl.c:2018-2230): the workspace is built from the seven stage blocks of an
        MPC problem (nominal values, replicated to every instance); `dims` = (N, nx, nu, ny, nt).  P blocks upper
        triangular.  The assembled patterns are kept as self.P / self.A (value order of update_P_A / update_recursive)."""
        from scipy import sparse
        L = _lib.lib()
        self = cls.__new__(cls)
        blocks = [CscPattern(sparse.triu(sparse.csc_matrix(b), format="csc")) for b in (Q0, Qi, QN)] + \
                 [CscPattern(sparse.csc_matrix(b)) for b in (A0, Ai, Aij, AN)]
        sd = _lib.StageDims(*[int(v) for v in dims])
        self.batch = int(q.shape[0])
        self.settings = default_settings(**settings)
        Pp, Ap = C.POINTER(_lib.Csc)(), C.POINTER(_lib.Csc)()
        self.h = C.c_void_p()
        self.status = _lib.check(L.osqp_batch_setup_recursive(C.byref(self.h), self.batch, C.byref(sd), *[b.ref for b in blocks],
                                                              _dptr(q), _dptr(l), _dptr(u), C.byref(self.settings), C.byref(Pp),
                                                              C.byref(Ap), None), "osqp_batch_setup_recursive")

        def to_scipy(M):
            n, m = M.n, M.m
            p = np.array([M.p[i] for i in range(n + 1)]); nz = int(p[-1])
            i = np.array([M.i[k] for k in range(nz)]); x = np.array([M.x[k] for k in range(nz)])
            return sparse.csc_matrix((x, i, p), shape=(m, n))
        P, A = to_scipy(Pp.contents), to_scipy(Ap.contents)
        L.rldl_csc_free(Pp); L.rldl_csc_free(Ap)
        self.P, self.A = CscPattern(P), CscPattern(A)
        self.n, self.m = self.P.shape[0], self.A.shape[0]
        self._device = q.device
        return self

    def update_recursive(self, first_stage, Px=None, Ax=None):
        """New values from stage `first_stage` on: the factorisation restarts there (LDL_update_from_pivot semantics)."""
        return int(_lib.lib().osqp_batch_update_recursive(self.h, int(first_stage), _dptr(Px), _dptr(Ax)))

    def partial_update_bounds(self, start, stop, l, u, wait=True):
        """osqp_partial_update_bounds (src/recursive_ldl.c:119-200): rows [start, stop) of the bounds of every instance.
        wait=False only enqueues (the l <= u verdict stays on the device and is raised by the next wait())."""
        _dev_f64(l, (self.batch, stop - start), "l"); _dev_f64(u, (self.batch, stop - start), "u")
        f = _lib.lib().osqp_batch_partial_update_bounds if wait else _lib.lib().osqp_batch_partial_update_bounds_async
        return int(f(self.h, int(start), int(stop), _dptr(l), _dptr(u)))

    def linsys(self):
        return BatchLinsys(self.P, self.A, None, None, 0, None, _handle=_lib.lib().osqp_batch_linsys(self.h), _owned=False)

    def solve(self, clone=True):
        """osqp_solve for every instance.  clone=False returns zero-copy views of the workspace's result arrays (valid
        until the next call that changes them) instead of a dozen device-to-device copies."""
        rc = _lib.lib().osqp_batch_solve(self.h)
        if rc:
            raise RuntimeError("osqp_batch_solve failed (%d)" % rc)
        return self.results(clone=clone)

    def solve_async(self):
        """Enqueue the solve on the workspace's stream (no host wait when the settings allow it, see the C header)."""
        rc = _lib.lib().osqp_batch_solve_async(self.h)
        if rc:
            raise RuntimeError("osqp_batch_solve_async failed (%d)" % rc)

    def wait(self, clone=True):
        rc = _lib.lib().osqp_batch_wait(self.h)
        if rc:
            raise RuntimeError("osqp_batch_wait failed (%d)" % rc)
        return self.results(clone=clone)

    def _view(self, ptr, shape, dtype, ld=None):
        """Zero-copy torch view of a workspace-owned device array (valid until cleanup).  ld: row stride in elements when the rows
        of the array are longer than shape[1] (single-store horizon handles)."""
        import torch
        count = int(np.prod(shape))
        if count == 0:
            return torch.empty(shape, dtype=dtype, device=self._device)
        itemsize = torch.empty((), dtype=dtype).element_size()

        class _Arr:  # __cuda_array_interface__ carrier
            pass
        a = _Arr()
        typestr = {torch.float64: "<f8", torch.int32: "<i4"}[dtype]
        strides = None if ld is None or len(shape) != 2 or ld == shape[1] else (int(ld) * itemsize, itemsize)
        a.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2,
                                      "strides": strides}
        return torch.as_tensor(a, device=self._device)

    def results(self, clone=True):
        import torch
        p = [C.c_void_p() for _ in range(8)]
        _lib.lib().osqp_batch_get(self.h, *[C.byref(t) for t in p])
        B, n, m = self.batch, self.n, self.m
        ldn, ldm = getattr(self, "_ldn", None), getattr(self, "_ldm", None)
        out = dict(x=self._view(p[0].value, (B, n), torch.float64, ldn), y=self._view(p[1].value, (B, m), torch.float64, ldm),
                   z=self._view(p[2].value, (B, m), torch.float64, ldm), status=self._view(p[3].value, (B,), torch.int32),
                   iter=self._view(p[4].value, (B,), torch.int32), obj=self._view(p[5].value, (B,), torch.float64),
                   pri_res=self._view(p[6].value, (B,), torch.float64), dua_res=self._view(p[7].value, (B,), torch.float64))
        it = [C.c_void_p() for _ in range(5)]
        _lib.lib().osqp_batch_get_iterates(self.h, *[C.byref(t) for t in it])
        out.update(x_iter=self._view(it[0].value, (B, n), torch.float64, ldn), y_iter=self._view(it[1].value, (B, m), torch.float64, ldm),
                   delta_x=self._view(it[3].value, (B, n), torch.float64, ldn), delta_y=self._view(it[4].value, (B, m), torch.float64, ldm))
        ps = C.c_void_p()
        _lib.lib().osqp_batch_get_polish_status(self.h, C.byref(ps))
        out["status_polish"] = self._view(ps.value, (B,), torch.int32)
        rh = [C.c_void_p() for _ in range(3)]
        _lib.lib().osqp_batch_get_rho(self.h, *[C.byref(t) for t in rh])
        out.update(rho=self._view(rh[0].value, (B,), torch.float64), rho_estimate=self._view(rh[1].value, (B,), torch.float64),
                   rho_updates=self._view(rh[2].value, (B,), torch.int32))
        if clone:
            out = {k: v.clone() for k, v in out.items()}
        return out

    def scaling_vectors(self):
        """(D, E, c) of the Ruiz equilibration (OSQPScaling), or None when settings.scaling == 0."""
        import torch
        p = [C.c_void_p() for _ in range(3)]
        if _lib.lib().osqp_batch_get_scaling(self.h, *[C.byref(t) for t in p]):
            return None
        return (self._view(p[0].value, (self.batch, self.n), torch.float64).clone(),
                self._view(p[1].value, (self.batch, self.m), torch.float64).clone(),
                self._view(p[2].value, (self.batch,), torch.float64).clone())

    def pack_results(self, rec):
        """osqp_batch_pack_results: the result records [x | y | obj | pri_res | dua_res | iter | status] of all instances into
        rec [batch, n + m + 5] (float64, device), one launch on the workspace's stream."""
        _dev_f64(rec, (self.batch, self.n + self.m + 5), "rec")
        rc = int(_lib.lib().osqp_batch_pack_results(self.h, _dptr(rec)))
        if rc:
            raise RuntimeError("osqp_batch_pack_results failed (%d)" % rc)
        return rec

    def update_lin_cost(self, q):
        return int(_lib.lib().osqp_batch_update_lin_cost(self.h, _dptr(_dev_f64(q, (self.batch, self.n), "q"))))

    def update_bounds(self, l, u, wait=True):
        """osqp_update_bounds; wait=False only enqueues: a refused update (l > u somewhere) changes nothing and is raised by the next wait()."""
        _dev_f64(l, (self.batch, self.m), "l"); _dev_f64(u, (self.batch, self.m), "u")
        f = _lib.lib().osqp_batch_update_bounds if wait else _lib.lib().osqp_batch_update_bounds_async
        return int(f(self.h, _dptr(l), _dptr(u)))

    def update_settings(self, **kw):
        """osqp_update_max_iter / _eps_* / _alpha / _warm_start / _scaled_termination / _check_termination /
        _polish_refine_iter / _delta: returns the C return code (1 = a value failed the reference's range check)."""
        for k, v in kw.items():
            if not hasattr(self.settings, k):
                raise KeyError("unknown setting %r" % k)
        new = type(self.settings)()
        C.memmove(C.byref(new), C.byref(self.settings), C.sizeof(new))
        for k, v in kw.items():
            setattr(new, k, v)
        rc = int(_lib.lib().osqp_batch_update_settings(self.h, C.byref(new)))
        if rc == 0:
            self.settings = new
        return rc

    def update_rho(self, rho):
        return int(_lib.lib().osqp_batch_update_rho(self.h, float(rho)))

    def update_P_A(self, Px=None, Ax=None, wait=True):
        """osqp_update_P_A.  wait=False only enqueues the scatter + refactorisation (no host synchronisation); a failed
        refactorisation then surfaces at the next wait() / solve()."""
        _dev_f64(Px, (self.batch, self.P.nnz), "Px"); _dev_f64(Ax, (self.batch, self.A.nnz), "Ax")
        fn = _lib.lib().osqp_batch_update_P_A if wait else _lib.lib().osqp_batch_update_P_A_async
        return int(fn(self.h, _dptr(Px), _dptr(Ax)))

    def warm_start(self, x, y):
        _dev_f64(x, (self.batch, self.n), "x"); _dev_f64(y, (self.batch, self.m), "y")
        return int(_lib.lib().osqp_batch_warm_start(self.h, _dptr(x), _dptr(y)))

    def time_iteration(self, reps=0):
        ms = _lib.c_float(0)
        if _lib.lib().osqp_batch_time_iteration(self.h, int(reps), C.byref(ms)):
            raise RuntimeError("time_iteration failed")
        return ms.value

    def last_loop(self):
        """(device ms of the last solve loop, ADMM iterations it ran, launch groups)."""
        ms, it, gr = _lib.c_float(0), _lib.c_int(0), _lib.c_int(0)
        if _lib.lib().osqp_batch_last_loop(self.h, C.byref(ms), C.byref(it), C.byref(gr)):
            raise RuntimeError("no solve loop has run")
        return ms.value, int(it.value), int(gr.value)

    def trace_iteration(self, iters=1):
        """Wave timeline of one launch of `iters` fused iterations: int64 array [batch, 8] (see include/osqp_rldl_hip.h)."""
        out = np.zeros((self.batch, 8), np.int64)
        if _lib.lib().osqp_batch_trace_iteration(self.h, int(iters), out.ctypes.data_as(C.c_void_p)):
            raise RuntimeError("trace_iteration failed")
        return out

    def cleanup(self):
        if getattr(self, "h", None) and getattr(self, "_owned", True):
            _lib.lib().osqp_batch_cleanup(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass


def _csc_to_scipy(M):
    from scipy import sparse
    n, m = M.n, M.m
    p = np.array([M.p[i] for i in range(n + 1)]); nz = int(p[-1])
    i = np.array([M.i[k] for k in range(nz)]); x = np.array([M.x[k] for k in range(nz)])
    return sparse.csc_matrix((x, i, p), shape=(m, n))


def _pad(M, shape):
    """M in the top-left corner of a zero matrix of the given shape"""
    from scipy import sparse
    M = sparse.coo_matrix(M)
    return sparse.csc_matrix((M.data, (M.row, M.col)), shape=shape)


class _HorizonStoreView(OSQPBatch):
    """The current horizon of a single-store OSQPHorizon seen as an OSQPBatch: n, m, P, A are the horizon's own; values and iterates
    go in through the horizon handle (packed arrays in the horizon's value order and sizes), results come out as strided views of
    the store's Nmax-sized rows."""

    def update_P_A(self, Px=None, Ax=None, wait=True):
        _dev_f64(Px, (self.batch, self.P.nnz), "Px"); _dev_f64(Ax, (self.batch, self.A.nnz), "Ax")
        return int(_lib.lib().osqp_horizon_update_P_A(self._hz.h, _dptr(Px), _dptr(Ax)))

    def warm_start(self, x, y):
        _dev_f64(x, (self.batch, self.n), "x"); _dev_f64(y, (self.batch, self.m), "y")
        return int(_lib.lib().osqp_horizon_warm_start(self._hz.h, _dptr(x), _dptr(y)))

    def linsys(self):
        P, A = self._hz.store_patterns()
        return BatchLinsys(CscPattern(P), CscPattern(A), None, None, 0, None, _handle=_lib.lib().osqp_batch_linsys(self.h), _owned=False)

    def update_lin_cost(self, q):
        raise NotImplementedError("single-store horizon: q, l, u are set by OSQPHorizon.update")

    update_bounds = update_lin_cost


class OSQPHorizon:
    """Variable-horizon MPC (osqp_setup_recursive with Nmax + osqp_update_recursive, src/recursive_ldl.c:2018-2230,
    :1973-2016): the batch is set up at horizon dims[0] and moves anywhere in 1..Nmax with `update`.  `workspace` is the
    OSQPBatch of the current horizon (owned by this object)."""

    def __init__(self, dims, Nmax, Q0, Qi, QN, A0, Ai, Aij, AN, q, l, u, **settings):
        from scipy import sparse
        L = _lib.lib()
        self._blocks = [CscPattern(sparse.triu(sparse.csc_matrix(b), format="csc")) for b in (Q0, Qi, QN)] + \
                       [CscPattern(sparse.csc_matrix(b)) for b in (A0, Ai, Aij, AN)]
        self.dims = tuple(int(v) for v in dims)
        self.Nmax = int(Nmax)
        self.batch = int(q.shape[0])
        self.settings = default_settings(**settings)
        self._device = q.device
        self._views = {}
        sd = _lib.StageDims(*self.dims)
        self.h = C.c_void_p()
        _lib.check(L.osqp_horizon_setup(C.byref(self.h), self.batch, C.byref(sd), self.Nmax, *[b.ref for b in self._blocks],
                                        _dptr(q), _dptr(l), _dptr(u), C.byref(self.settings), None), "osqp_horizon_setup")

    @property
    def N(self):
        return int(_lib.lib().osqp_horizon_N(self.h))

    def sizes(self, N=None):
        """(n, m) of horizon N (default: the current one)."""
        N = self.N if N is None else int(N)
        _, nx, nu, ny, nt = self.dims
        return N * (nx + nu), N * (nx + ny) + nt

    def patterns(self, N):
        """Assembled P (upper triangular) and A of horizon N with the nominal values: the value order of update_P_A."""
        L = _lib.lib()
        sd = _lib.StageDims(int(N), *self.dims[1:])
        Pp, Ap = C.POINTER(_lib.Csc)(), C.POINTER(_lib.Csc)()
        _lib.check(L.rldl_setup_AP_matrices(C.byref(sd), *[b.ref for b in self._blocks], C.byref(Pp), C.byref(Ap),
                                            None, None, None, None, None, None), "rldl_setup_AP_matrices")
        P, A = _csc_to_scipy(Pp.contents), _csc_to_scipy(Ap.contents)
        L.rldl_csc_free(Pp); L.rldl_csc_free(Ap)
        return P, A

    @property
    def single_store(self):
        """True when ONE workspace at Nmax dimensions serves every horizon (scaling = 0, product tri-solve available)."""
        return bool(_lib.lib().osqp_horizon_is_single(self.h))

    @property
    def n_workspaces(self):
        """Resident numeric workspaces: 1 with the single store, one per visited horizon otherwise."""
        return int(_lib.lib().osqp_horizon_workspaces(self.h))

    def store_patterns(self):
        """Single store: the assembled P / A patterns of the one workspace (Nmax stages, interior blocks = the unions Qi + QN on
        the state part and Ai + AN in the first nt rows)."""
        from scipy import sparse
        L = _lib.lib()
        _, nx, nu, ny, nt = self.dims
        Q0, Qi, QN, A0, Ai, Aij, AN = [sparse.csc_matrix((np.ones(b.nnz), b.i, b.p), shape=b.shape) for b in self._blocks]

        def struct(M):
            M = sparse.csc_matrix(M).copy(); M.data[:] = 1.0; return M
        Qi_u = sparse.triu(sparse.csc_matrix(struct(Qi) + _pad(struct(QN), Qi.shape)), format="csc")
        Ai_u = sparse.csc_matrix(struct(Ai) + _pad(struct(AN), Ai.shape))
        blocks = [CscPattern(sparse.triu(struct(Q0), format="csc")), CscPattern(Qi_u), CscPattern(sparse.triu(struct(QN), format="csc")),
                  CscPattern(struct(A0)), CscPattern(Ai_u), CscPattern(struct(Aij)), CscPattern(struct(AN))]
        sd = _lib.StageDims(self.Nmax, nx, nu, ny, nt)
        Pp, Ap = C.POINTER(_lib.Csc)(), C.POINTER(_lib.Csc)()
        _lib.check(L.rldl_setup_AP_matrices(C.byref(sd), *[b.ref for b in blocks], C.byref(Pp), C.byref(Ap),
                                            None, None, None, None, None, None), "rldl_setup_AP_matrices")
        P, A = _csc_to_scipy(Pp.contents), _csc_to_scipy(Ap.contents)
        L.rldl_csc_free(Pp); L.rldl_csc_free(Ap)
        return P, A

    @property
    def workspace(self):
        N = self.N
        if N not in self._views:
            single = self.single_store
            w = (_HorizonStoreView if single else OSQPBatch).__new__(_HorizonStoreView if single else OSQPBatch)
            w._owned = False
            w._stream = None
            w.h = C.c_void_p(_lib.lib().osqp_horizon_workspace(self.h))
            P, A = self.patterns(N)
            w.P, w.A = CscPattern(P), CscPattern(A)
            w.n, w.m = w.P.shape[0], w.A.shape[0]
            w.batch, w._device = self.batch, self._device
            w.settings = default_settings()
            C.memmove(C.byref(w.settings), C.byref(self.settings), C.sizeof(self.settings))
            w.status = 0
            if single:
                a, b = _lib.c_int(0), _lib.c_int(0)
                _lib.lib().osqp_horizon_ld(self.h, C.byref(a), C.byref(b))
                w._ldn, w._ldm, w._hz = int(a.value), int(b.value), self
            self._views[N] = w
        return self._views[N]

    def update(self, N, q, l, u):
        """osqp_update_recursive: move to horizon N with the new problem's q, l, u.  Returns the C return code
        (-1: N outside 1..Nmax, as the reference)."""
        n, m = self.sizes(N)
        if 1 <= int(N) <= self.Nmax:
            _dev_f64(q, (self.batch, n), "q"); _dev_f64(l, (self.batch, m), "l"); _dev_f64(u, (self.batch, m), "u")
        return int(_lib.lib().osqp_horizon_update(self.h, int(N), _dptr(q), _dptr(l), _dptr(u)))

    def last_update(self):
        a, b, c = _lib.c_int(0), _lib.c_int(0), _lib.c_int(0)
        _lib.lib().osqp_horizon_last_update(self.h, C.byref(a), C.byref(b), C.byref(c))
        return dict(pivot_stage=int(a.value), instances_reused=int(b.value), workspace_created=bool(c.value))

    def free(self):
        if getattr(self, "h", None):
            for w in self._views.values():
                w.h = C.c_void_p()
            _lib.lib().osqp_horizon_free(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

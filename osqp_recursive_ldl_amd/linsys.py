"""Host-side mirror of the reference's linear-system plugin interface, over the C-ABI.

`HipLDLSolver`      legacy single-instance object: init / solve / update_matrices / update_rho_vec / free with
                    HOST numpy data, exactly the call shapes of lin_sys/direct/qdldl/qdldl_interface.c
                    (init :170-316, solve :559-585, update_matrices :590-602, update_rho_vec :605-619).
`BatchLinsys`       the same five operations with a leading batch dimension on DEVICE tensors.

torch is used only as the owner of device memory and the current stream; all compute happens inside
libosqp_rldl_hip.so.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Csc, c_int


def _ip(a):
    return a.ctypes.data_as(_lib.IP)


def _fp(a):
    return a.ctypes.data_as(_lib.FP)


class CscPattern:
    """A `csc` struct (include/types.h:21-29) backed by numpy arrays kept alive by this object."""

    def __init__(self, M, upper=False):
        from scipy import sparse
        M = sparse.csc_matrix(M)
        if upper:
            M = sparse.triu(M, format="csc")
        M.sort_indices()
        self.shape = M.shape
        self.p = np.ascontiguousarray(M.indptr, dtype=np.int64)
        self.i = np.ascontiguousarray(M.indices, dtype=np.int64)
        self.x = np.ascontiguousarray(M.data, dtype=np.float64)
        self.nnz = int(self.p[-1])
        self.s = Csc(max(self.nnz, 1), M.shape[0], M.shape[1], _ip(self.p), _ip(self.i), _fp(self.x), -1)

    @property
    def ref(self):
        return C.byref(self.s)

    def with_values(self, x):
        out = CscPattern.__new__(CscPattern)
        out.shape, out.p, out.i, out.nnz = self.shape, self.p, self.i, self.nnz
        out.x = np.ascontiguousarray(x, dtype=np.float64)
        assert out.x.shape == (self.nnz,)
        out.s = Csc(max(self.nnz, 1), self.shape[0], self.shape[1], _ip(out.p), _ip(out.i), _fp(out.x), -1)
        return out


class HipLDLSolver:
    """Legacy single-instance plugin object (one QP, host arrays), driven through the vtable of the
    returned struct -- the same way OSQP's `work->linsys_solver->solve(...)` would (src/auxil.c:185)."""

    def __init__(self, P, A, sigma, rho_vec, polish=0):
        L = _lib.lib()
        self.P = P if isinstance(P, CscPattern) else CscPattern(P)
        self.A = A if isinstance(A, CscPattern) else CscPattern(A)
        self.n, self.m = self.P.shape[0], self.A.shape[0]
        self._sp = C.POINTER(_lib.HipldlSolver)()
        rv = None if rho_vec is None else np.ascontiguousarray(rho_vec, dtype=np.float64)
        self.status = _lib.check(L.init_linsys_solver_hipldl(C.byref(self._sp), self.P.ref, self.A.ref, float(sigma),
                                                             None if rv is None else _fp(rv), int(polish)),
                                 "init_linsys_solver_hipldl")
        self.type = self._sp.contents.type if self.status == 0 else None

    def solve(self, b):
        b = np.array(b, dtype=np.float64, copy=True)
        rc = self._sp.contents.solve(self._sp, _fp(b))     # through the vtable
        if rc:
            raise RuntimeError("solve failed")
        return b

    def update_matrices(self, P, A):
        Pn = self.P.with_values(P if not hasattr(P, "data") else _sorted_data(P))
        An = self.A.with_values(A if not hasattr(A, "data") else _sorted_data(A))
        return int(self._sp.contents.update_matrices(self._sp, Pn.ref, An.ref))

    def update_rho_vec(self, rho_vec):
        rv = np.ascontiguousarray(rho_vec, dtype=np.float64)
        return int(self._sp.contents.update_rho_vec(self._sp, _fp(rv)))

    def free(self):
        if self._sp:
            self._sp.contents.free(self._sp)
            self._sp = C.POINTER(_lib.HipldlSolver)()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _sorted_data(M):
    from scipy import sparse
    M = sparse.csc_matrix(M)
    M.sort_indices()
    return M.data


def _dptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _dev_f64(t, shape, name):
    import torch
    if t is None:
        return None
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
        raise TypeError("%s must be a contiguous float64 device tensor" % name)
    if tuple(t.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return t


class BatchLinsys:
    """Batched plugin: `batch` instances sharing the sparsity pattern of P (upper triangular) and A.
    Value tensors are float64 device tensors, instance-major: Px[batch, nnzP], Ax[batch, nnzA],
    rho_vec[batch, m], b[batch, n+m]."""

    def __init__(self, P_pattern, A_pattern, Px, Ax, sigma, rho_vec, polish=0, perm=None, _handle=None, _owned=True):
        L = _lib.lib()
        self.P = P_pattern if isinstance(P_pattern, CscPattern) else CscPattern(P_pattern)
        self.A = A_pattern if isinstance(A_pattern, CscPattern) else CscPattern(A_pattern)
        self.n, self.m = self.P.shape[0], self.A.shape[0]
        self._owned = _owned
        if _handle is not None:
            self.h = C.c_void_p(_handle)
            self.batch = self.dims()["batch"]
            self.status = 0
            return
        self.batch = int(Px.shape[0])
        _dev_f64(Px, (self.batch, self.P.nnz), "Px"); _dev_f64(Ax, (self.batch, self.A.nnz), "Ax")
        if not polish:
            _dev_f64(rho_vec, (self.batch, self.m), "rho_vec")
        pm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int64)
        self.h = C.c_void_p()
        self.status = _lib.check(L.rldl_batch_init(C.byref(self.h), self.batch, self.P.ref, self.A.ref, _dptr(Px), _dptr(Ax),
                                                   float(sigma), None if polish else _dptr(rho_vec), int(polish),
                                                   None if pm is None else _ip(pm), None), "rldl_batch_init")

    @classmethod
    def recursive(cls, dims, P_pattern, A_pattern, Px, Ax, sigma, rho_vec):
        """Stage-recursive strategy (src/recursive_ldl.c): dims = (N, nx, nu, ny, nt)."""
        L = _lib.lib()
        self = cls.__new__(cls)
        self.P = P_pattern if isinstance(P_pattern, CscPattern) else CscPattern(P_pattern)
        self.A = A_pattern if isinstance(A_pattern, CscPattern) else CscPattern(A_pattern)
        self.n, self.m = self.P.shape[0], self.A.shape[0]
        self.batch = int(Px.shape[0])
        self._owned = True
        sd = _lib.StageDims(*[int(v) for v in dims])
        self.h = C.c_void_p()
        self.status = _lib.check(L.rldl_batch_init_recursive(C.byref(self.h), self.batch, C.byref(sd), self.P.ref, self.A.ref,
                                                             _dptr(Px), _dptr(Ax), float(sigma), _dptr(rho_vec), None),
                                 "rldl_batch_init_recursive")
        return self

    def solve(self, b):
        """In place on b[batch, n+m] (qdldl_interface.c:559-585)."""
        _dev_f64(b, (self.batch, self.n + self.m), "b")
        if _lib.lib().rldl_batch_solve(self.h, _dptr(b)):
            raise RuntimeError("rldl_batch_solve failed")
        return b

    def update_matrices(self, Px=None, Ax=None):
        _dev_f64(Px, (self.batch, self.P.nnz), "Px"); _dev_f64(Ax, (self.batch, self.A.nnz), "Ax")
        return int(_lib.lib().rldl_batch_update_matrices(self.h, _dptr(Px), _dptr(Ax)))

    def update_rho_vec(self, rho_vec, mask=None):
        _dev_f64(rho_vec, (self.batch, self.m), "rho_vec")
        return int(_lib.lib().rldl_batch_update_rho_vec(self.h, _dptr(rho_vec), _dptr(mask)))

    def update_from_stage(self, first_stage, Px=None, Ax=None, rho_vec=None):
        return int(_lib.lib().rldl_batch_update_from_stage(self.h, int(first_stage), _dptr(Px), _dptr(Ax), _dptr(rho_vec)))

    def dims(self):
        v = [c_int(0) for _ in range(5)]
        _lib.lib().rldl_batch_dims(self.h, *[C.byref(t) for t in v])
        return dict(n=v[0].value, m=v[1].value, nnzKKT=v[2].value, nnzL=v[3].value, batch=v[4].value)

    def export_symbolic(self):
        d = self.dims()
        N = d["n"] + d["m"]
        out = dict(perm=np.zeros(N, np.int64), etree=np.zeros(N, np.int64), Lnz=np.zeros(N, np.int64),
                   Lp=np.zeros(N + 1, np.int64), Li=np.zeros(max(d["nnzL"], 1), np.int64),
                   KKTp=np.zeros(N + 1, np.int64), KKTi=np.zeros(d["nnzKKT"], np.int64),
                   PtoKKT=np.zeros(max(self.P.nnz, 1), np.int64), AtoKKT=np.zeros(max(self.A.nnz, 1), np.int64),
                   rhotoKKT=np.zeros(max(d["m"], 1), np.int64))
        _lib.lib().rldl_batch_export_symbolic(self.h, *[_ip(out[k]) for k in
                                                        ("perm", "etree", "Lnz", "Lp", "Li", "KKTp", "KKTi", "PtoKKT",
                                                         "AtoKKT", "rhotoKKT")])
        out["Li"] = out["Li"][:d["nnzL"]]
        out["PtoKKT"] = out["PtoKKT"][:self.P.nnz]; out["AtoKKT"] = out["AtoKKT"][:self.A.nnz]
        out["rhotoKKT"] = out["rhotoKKT"][:d["m"]]
        return out

    def export_factor(self, inst):
        d = self.dims()
        N = d["n"] + d["m"]
        Lx = np.zeros(max(d["nnzL"], 1)); D = np.zeros(N); Dinv = np.zeros(N); Kx = np.zeros(d["nnzKKT"])
        if _lib.lib().rldl_batch_export_factor(self.h, int(inst), _fp(Lx), _fp(D), _fp(Dinv), _fp(Kx)):
            raise RuntimeError("export_factor failed")
        return dict(Lx=Lx[:d["nnzL"]], D=D, Dinv=Dinv, KKTx=Kx)

    def export_prod(self, inst=0):
        """Stage handles: tables of the product tri-solve and the tile values of one instance (None when the handle has none)."""
        L = _lib.lib()
        meta = np.zeros(8, np.int64)
        if L.rldl_batch_export_prod(self.h, int(inst), _ip(meta), None, None, None, None, None, None):
            return None
        nt, nw, nTi, nb, ng = int(meta[1]), int(meta[2]), int(meta[3]), int(meta[4]), int(meta[7])
        prog = np.zeros(12 * max(ng, 1), np.int32); tinfo = np.zeros(4 * (nt + 1), np.int32); tab = np.zeros(max(nw, 1), np.uint32)
        src = np.zeros(max(nTi, 1), np.uint16); blk = np.zeros(2 * nb, np.int32); Ti = np.zeros(max(nTi, 1))
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        if L.rldl_batch_export_prod(self.h, int(inst), _ip(meta), vp(prog), vp(tinfo), vp(tab), vp(src), vp(blk), _fp(Ti)):
            raise RuntimeError("export_prod failed")
        return dict(tiles=nt, steps=ng, nb=nb, ld=int(meta[5]), kmax=int(meta[6]), prog=prog.reshape(-1, 12), tinfo=tinfo.reshape(-1, 4)[:nt],
                    tab=tab[:nw], src=src[:nTi], blk=blk.reshape(-1, 2), Ti=Ti[:nTi], mode=int(meta[0]))

    def factor_status(self):
        st = np.zeros(self.batch, np.int64)
        _lib.lib().rldl_batch_factor_status(self.h, _ip(st))
        return st

    def time_solve(self, b, reps=20):
        ms = _lib.c_float(0)
        if _lib.lib().rldl_batch_time_solve(self.h, _dptr(b), int(reps), C.byref(ms)):
            raise RuntimeError("time_solve failed")
        return ms.value

    @staticmethod
    def time_solve_rotating(handles, bs, reps=200):
        """ms per launch of the solve kernel over a rotation of handles (same stream) and right-hand sides: with a combined working
        set beyond the Infinity Cache every launch streams its factor rows from HBM."""
        ms = _lib.c_float(0)
        hs = (C.c_void_p * len(handles))(*[h.h for h in handles])
        ps = (C.c_void_p * len(bs))(*[C.c_void_p(b.data_ptr()) for b in bs])
        if _lib.lib().rldl_batch_time_solve_rotating(hs, ps, len(handles), int(reps), C.byref(ms)):
            raise RuntimeError("time_solve_rotating failed")
        return ms.value

    def trace_solve(self, b):
        """Wave timeline of one launch of the solve kernel: int64 [batch, 8] ticks of the 100 MHz device clock (wave start, loads
        landed, forward gather / forward product / backward product / scatter done, stores issued, 0); None when the handle's kernel carries no timeline.  Solves b in place."""
        out = np.zeros((self.batch, 8), np.int64)
        rc = _lib.lib().rldl_batch_trace_solve(self.h, _dptr(b), out.ctypes.data_as(C.c_void_p))
        if rc == 2:
            return None
        if rc:
            raise RuntimeError("trace_solve failed")
        return out

    def set_cache_policy(self, policy):
        """rldl_batch_set_cache_policy: "auto" (default), "resident" (one factorisation, many solves: keep the rows in the Infinity
        Cache) or "stream" (many handles in turn / rows beyond the cache: non-temporal loads)."""
        if _lib.lib().rldl_batch_set_cache_policy(self.h, {"auto": 0, "resident": 1, "stream": 2}[policy]):
            raise RuntimeError("set_cache_policy failed")

    def trace_factor(self):
        """Wave timeline of one numeric factorisation of the values the handle holds: int64 [batch, 8] ticks of the 100 MHz device
        clock (wave start, KKT values in the workspace, head contributions added, tail in registers, tail eliminated, factor row
        stored, triangle packed, tail inverse stored); None when the pattern is not served by the arrowhead kernel."""
        out = np.zeros((self.batch, 8), np.int64)
        rc = _lib.lib().rldl_batch_trace_factor(self.h, out.ctypes.data_as(C.c_void_p))
        if rc == 2:
            return None
        if rc:
            raise RuntimeError("trace_factor failed")
        return out

    def free(self):
        if getattr(self, "h", None) and self._owned:
            _lib.lib().rldl_batch_free(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def symbolic_analyze(P, A, polish=0, perm=None):
    """Host-only symbolic phase (no GPU needed): permutation, permuted KKT pattern, scatter maps,
    elimination tree and the pattern of L, as plain numpy int64 arrays."""
    L = _lib.lib()
    Pc = P if isinstance(P, CscPattern) else CscPattern(P)
    Ac = A if isinstance(A, CscPattern) else CscPattern(A)
    n, m = Pc.shape[0], Ac.shape[0]
    N = n + m
    pm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int64)
    nk, nl, h = c_int(0), c_int(0), c_int(0)
    null = [None] * 10
    rc = L.rldl_symbolic_analyze(Pc.ref, Ac.ref, int(polish), None if pm is None else _ip(pm), C.byref(nk), C.byref(nl),
                                 C.byref(h), *null)
    if rc:
        raise ValueError("rldl_symbolic_analyze failed (%d)" % rc)
    out = dict(perm=np.zeros(N, np.int64), etree=np.zeros(N, np.int64), Lnz=np.zeros(N, np.int64),
               Lp=np.zeros(N + 1, np.int64), Li=np.zeros(max(nl.value, 1), np.int64), KKTp=np.zeros(N + 1, np.int64),
               KKTi=np.zeros(max(nk.value, 1), np.int64), PtoKKT=np.zeros(max(Pc.nnz, 1), np.int64),
               AtoKKT=np.zeros(max(Ac.nnz, 1), np.int64), rhotoKKT=np.zeros(max(m, 1), np.int64))
    keys = ("perm", "etree", "Lnz", "Lp", "Li", "KKTp", "KKTi", "PtoKKT", "AtoKKT", "rhotoKKT")
    L.rldl_symbolic_analyze(Pc.ref, Ac.ref, int(polish), None if pm is None else _ip(pm), C.byref(nk), C.byref(nl),
                            C.byref(h), *[_ip(out[k]) for k in keys])
    out["Li"] = out["Li"][:nl.value]; out["KKTi"] = out["KKTi"][:nk.value]
    out["PtoKKT"] = out["PtoKKT"][:Pc.nnz]; out["AtoKKT"] = out["AtoKKT"][:Ac.nnz]; out["rhotoKKT"] = out["rhotoKKT"][:m]
    out.update(nnzKKT=nk.value, nnzL=nl.value, etree_height=h.value, n=n, m=m)
    return out


def plan_export(P, A, polish=0, perm=None):
    """Host-only export of the grouped solve plan (see csrc/rldl_plan.c): dict of numpy arrays."""
    L = _lib.lib()
    Pc = P if isinstance(P, CscPattern) else CscPattern(P)
    Ac = A if isinstance(A, CscPattern) else CscPattern(A)
    pm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int64)
    meta = np.zeros(48, np.int64)
    if L.rldl_plan_export(Pc.ref, Ac.ref, int(polish), None if pm is None else _ip(pm), _ip(meta), None, 0, None):
        raise ValueError("rldl_plan_export failed")
    words, nnzL = int(meta[4]), int(meta[20])
    blob = np.zeros(max(words, 1), np.int32)
    LtoS = np.zeros(max(nnzL, 1), np.int64)
    L.rldl_plan_export(Pc.ref, Ac.ref, int(polish), None if pm is None else _ip(pm), _ip(meta),
                       blob.ctypes.data_as(C.POINTER(C.c_int)), words, _ip(LtoS))
    names = ["plan_ok", "nS", "nO", "ngroups", "plan_words", "po_gstart", "po_gflag", "po_gToff", "po_fsp", "po_bsp", "po_fsb",
             "po_fsc", "po_bsb", "po_bsc", "po_fsig", "po_bsig", "po_fcol", "po_brs", "po_perm", "N", "nnzL", "arrow_ok", "arrow_group",
             "arrow_vsteps", "arrow_vrows", "po_avmap", "po_avcol", "po_avrow", "nOp", "tile_ok", "tile_ta", "tile_tq", "tile_lanes", "nTi",
             "po_tlane", "po_tmap", "po_tislot", "tile_admm_ok", "tile_vslots", "tile_slots", "po_tpos", "tile_ck0", "tile_ck1",
             "tile_ck2", "tile_tk", "po_cmap", "po_crow", "tile_sp"]
    out = {k: int(meta[i]) for i, k in enumerate(names)}
    out["blob"] = blob[:words]
    out["LtoS"] = LtoS[:nnzL]
    return out


def stage_prod_export(dims, P, A):
    """Host-only (no GPU): tables of the product tri-solve of a stage-structured pattern, or None when it does not qualify."""
    L = _lib.lib()
    Pc = P if isinstance(P, CscPattern) else CscPattern(P)
    Ac = A if isinstance(A, CscPattern) else CscPattern(A)
    sd = _lib.StageDims(*[int(v) for v in dims])
    meta = np.zeros(8, np.int64)
    if L.rldl_stage_prod_export(Pc.ref, Ac.ref, C.byref(sd), _ip(meta), None, None, None, None, None, None):
        return None
    nt, nw, nTi, nb, ns = int(meta[1]), int(meta[2]), int(meta[3]), int(meta[4]), int(meta[7])
    sym = symbolic_analyze(Pc, Ac, perm=__import__("osqp_recursive_ldl_amd").workloads.stage_permutation(*dims))
    prog = np.zeros(12 * max(ns, 1), np.int32); tinfo = np.zeros(4 * (nt + 1), np.int32); tab = np.zeros(max(nw, 1), np.uint32)
    src = np.zeros(max(nTi, 1), np.uint16); blk = np.zeros(2 * nb, np.int32); LtoS = np.zeros(max(sym["nnzL"], 1), np.int64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    if L.rldl_stage_prod_export(Pc.ref, Ac.ref, C.byref(sd), _ip(meta), vp(prog), vp(tinfo), vp(tab), vp(src), vp(blk), _ip(LtoS)):
        raise RuntimeError("rldl_stage_prod_export failed")
    return dict(tiles=nt, steps=ns, nb=nb, ld=int(meta[5]), kmax=int(meta[6]), prog=prog.reshape(-1, 12), tinfo=tinfo.reshape(-1, 4)[:nt],
                tab=tab[:nw], src=src[:nTi], blk=blk.reshape(-1, 2), LtoS=LtoS[:sym["nnzL"]], sym=sym)


def plan_emulate_solve(plan, S, Dinv, x):
    """CPU emulation of the device schedule (plan_tri_solve in csrc/rldl_kernels.hip), one instance:
    S = factor in plan slot order [nS], Dinv [N], x = permuted right-hand side [N] -> L^-T D^-1 L^-1 x."""
    w = plan["blob"]
    u16 = lambda off: w[off:].view(np.uint16)
    fsig, bsig, fcol = u16(plan["po_fsig"]), u16(plan["po_bsig"]), u16(plan["po_fcol"])
    xs = np.array(x, float)
    ng = plan["ngroups"]
    gs = w[plan["po_gstart"]:plan["po_gstart"] + ng + 1]
    for k in range(ng):                                   # forward
        g0, g = int(gs[k]), int(gs[k + 1] - gs[k])
        fs0, fs1 = int(w[plan["po_fsp"] + k]), int(w[plan["po_fsp"] + k + 1])
        if fs1 > fs0:
            ga = np.array([xs[fsig[g0 + i]] for i in range(g)])
            for t in range(fs0, fs1):
                base, cnt = int(w[plan["po_fsb"] + t]), int(w[plan["po_fsc"] + t])
                for i in range(cnt):
                    ga[i] -= S[base + i] * xs[fcol[base + i]]
            for i in range(g):
                xs[fsig[g0 + i]] = ga[i]
        if w[plan["po_gflag"] + k]:
            Tb = int(w[plan["po_gToff"] + k])
            acc = xs[g0:g0 + g].copy()
            for a in range(g - 1):
                for i in range(a + 1, g):
                    acc[i] -= S[Tb + i * (i - 1) // 2 + a] * acc[a]
            xs[g0:g0 + g] = acc
    for k in range(ng - 1, -1, -1):                       # backward
        g0, g = int(gs[k]), int(gs[k + 1] - gs[k])
        bs0, bs1 = int(w[plan["po_bsp"] + k]), int(w[plan["po_bsp"] + k + 1])
        scaled = False
        if bs1 > bs0:
            gb = np.array([xs[bsig[g0 + i]] * Dinv[bsig[g0 + i]] for i in range(g)])
            for t in range(bs0, bs1):
                base, cnt = int(w[plan["po_bsb"] + t]), int(w[plan["po_bsc"] + t])
                for i in range(cnt):
                    rs = int(w[plan["po_brs"] + base + i]) & 0xffffffff
                    gb[i] -= S[rs >> 16] * xs[rs & 0xffff]
            for i in range(g):
                xs[bsig[g0 + i]] = gb[i]
            scaled = True
        acc = xs[g0:g0 + g].copy() if scaled else xs[g0:g0 + g] * Dinv[g0:g0 + g]
        if w[plan["po_gflag"] + k]:
            Tb = int(w[plan["po_gToff"] + k])
            for il in range(g - 1, 0, -1):
                for j in range(il):
                    acc[j] -= S[Tb + il * (il - 1) // 2 + j] * acc[il]
        xs[g0:g0 + g] = acc
    return xs

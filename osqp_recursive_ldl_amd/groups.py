"""Batches whose instances do NOT share one sparsity pattern ("per-instance pattern" variant of config 2).

The backend factorises one pattern per handle, so the instances are bucketed by pattern: one OSQPBatch per distinct
(P pattern, A pattern).  With a fixed number of iterations (no termination checks, no rho adaptation, no polish) and patterns
that select the same kernel instantiation, a solve of ALL groups is three launches over the stacked instances plus one that
writes the results in the caller's instance order (osqp_multi_*, include/osqp_rldl_hip.h).  Otherwise every group is enqueued
on its own HIP stream before any is waited for, and the results are gathered with torch.  All instances must have the same
(n, m).
"""
import ctypes as C
import numpy as np

from . import _lib
from .linsys import CscPattern
from .osqp_batch import OSQPBatch


def _key(P, A):
    return (P.shape, A.shape, P.indptr.tobytes(), P.indices.tobytes(), A.indptr.tobytes(), A.indices.tobytes())


class OSQPBatchGroups:
    def __init__(self, problems, device="cuda:0", one_launch=True, **settings):
        """problems: sequence of (P, q, A, l, u) with scipy sparse P (upper triangular part is used) and A, numpy q, l, u.
        one_launch=False keeps every group on its own stream even when the set qualifies for the single launch chain."""
        import torch
        from scipy import sparse
        self.count = len(problems)
        buckets = {}
        canon = []
        for i, (P, q, A, l, u) in enumerate(problems):
            Pu = sparse.triu(sparse.csc_matrix(P), format="csc"); Pu.sort_indices()
            Ac = sparse.csc_matrix(A); Ac.sort_indices()
            canon.append((Pu, Ac))
            buckets.setdefault(_key(Pu, Ac), []).append(i)
        self.n, self.m = canon[0][0].shape[0], canon[0][1].shape[0]
        if any(c[0].shape[0] != self.n or c[1].shape[0] != self.m for c in canon):
            raise ValueError("all instances must have the same (n, m)")
        dev = torch.device(device)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        self.groups = []
        for idx in buckets.values():
            Pu, Ac = canon[idx[0]]
            Px = np.stack([canon[i][0].data for i in idx]); Ax = np.stack([canon[i][1].data for i in idx])
            q = np.stack([np.asarray(problems[i][1], float) for i in idx])
            l = np.stack([np.asarray(problems[i][3], float) for i in idx]); u = np.stack([np.asarray(problems[i][4], float) for i in idx])
            stream = torch.cuda.Stream(device=dev)
            torch.cuda.synchronize(dev)                              # uploads above ran on torch's current stream
            w = OSQPBatch(CscPattern(Pu), CscPattern(Ac), t(Px), t(Ax), t(q), t(l), t(u), stream=stream, **settings)
            if w.status != 0:
                raise RuntimeError("setup of a pattern group failed (%d)" % w.status)
            self.groups.append((torch.as_tensor(np.asarray(idx), device=dev), w))
        self._dev = dev
        # position of every original instance in the concatenation of the groups' result arrays
        order = np.concatenate([np.asarray(idx) for idx in buckets.values()])
        self._inv = torch.as_tensor(np.argsort(order, kind="stable"), device=dev)
        # all groups in one launch chain when the set qualifies (osqp_multi_create returns 2 otherwise)
        self._multi = None
        self._mstream = torch.cuda.Stream(device=dev)
        hs = (C.c_void_p * len(self.groups))(*[w.h for _, w in self.groups])
        dest = np.ascontiguousarray(order, dtype=np.int64)
        mh = C.c_void_p()
        rc = _lib.lib().osqp_multi_create(C.byref(mh), hs, len(self.groups), dest.ctypes.data_as(_lib.IP), C.c_void_p(self._mstream.cuda_stream)) if one_launch else 2
        if rc == 0:
            self._multi = mh
            B, n, m = self.count, self.n, self.m
            f = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
            i = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)
            self._out = dict(x=f(B, n), y=f(B, m), z=f(B, m), status=i(B), iter=i(B), obj=f(B), pri_res=f(B), dua_res=f(B))
        elif rc != 2:
            raise RuntimeError("osqp_multi_create failed (%d)" % rc)

    @property
    def n_patterns(self):
        return len(self.groups)

    @property
    def one_launch(self):
        """True when a solve of all groups is one launch chain (osqp_multi_solve)."""
        return self._multi is not None

    def solve(self):
        """Solve every group and return results in the original instance order (the arrays are reused by the next solve)."""
        import torch
        if self._multi is not None:
            L = _lib.lib()
            rc = L.osqp_multi_solve(self._multi)
            if rc == 2:                                      # a member's settings left the fixed-iteration case: per-workspace route from now on
                L.osqp_multi_free(self._multi)
                self._multi = None
                return self.solve()
            if rc:
                raise RuntimeError("osqp_multi_solve failed (%d)" % rc)
            o = self._out
            p = lambda t: C.c_void_p(t.data_ptr())
            if L.osqp_multi_get(self._multi, p(o["x"]), p(o["y"]), p(o["z"]), p(o["status"]), p(o["iter"]), p(o["obj"]), p(o["pri_res"]), p(o["dua_res"])):
                raise RuntimeError("osqp_multi_get failed")
            return dict(o)
        for _, w in self.groups:
            w.solve_async()
        parts, err = [], None
        for _, w in self.groups:                             # wait for every group before raising: a verdict must not linger in another group
            try:
                parts.append(w.wait(clone=False))
            except RuntimeError as e:
                err = e
        if err is not None:
            raise err
        # one concatenation + one gather per result field (not one indexed copy per field and group)
        return {key: torch.cat([res[key] for res in parts], 0)[self._inv] for key in parts[0]}

    def update_P_A(self, values):
        """New P / A values for every group: values[k] = (Px, Ax) device tensors [batch_k, nnz] of group k (the order of self.groups).
        One launch chain over all groups when the set qualifies, else one asynchronous update per workspace; a failed
        refactorisation surfaces at the next solve."""
        import torch
        if len(values) != len(self.groups):
            raise ValueError("one (Px, Ax) pair per pattern group")
        # the values were produced on torch's current stream; the chain runs on other streams: order them behind it
        # (an event on the producer stream) and tell the caching allocator which streams still read the arrays
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self._dev))
        if self._multi is not None:
            self._mstream.wait_event(ready)
            for v in values:
                v[0].record_stream(self._mstream); v[1].record_stream(self._mstream)
            G = len(values)
            px = (C.c_void_p * G)(*[C.c_void_p(v[0].data_ptr()) for v in values])
            ax = (C.c_void_p * G)(*[C.c_void_p(v[1].data_ptr()) for v in values])
            rc = _lib.lib().osqp_multi_update_P_A(self._multi, px, ax)
            if rc == 0:
                return
            if rc != 2:
                raise RuntimeError("osqp_multi_update_P_A failed (%d)" % rc)
        for (_, w), (Px, Ax) in zip(self.groups, values):
            if w._stream is not None:
                w._stream.wait_event(ready)
                Px.record_stream(w._stream); Ax.record_stream(w._stream)
            if w.update_P_A(Px, Ax, wait=False):
                raise RuntimeError("update_P_A failed")

    def cleanup(self):
        if self._multi is not None:
            _lib.lib().osqp_multi_free(self._multi)
            self._multi = None
        for _, w in self.groups:
            w.cleanup()
        self.groups = []

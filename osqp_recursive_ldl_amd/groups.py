"""Batches whose instances do NOT share one sparsity pattern ("per-instance pattern" variant of config 2) -- down to
one pattern per instance (the reference's semantics: every osqp_setup owns its pattern, qdldl_interface.c:99-166).

The backend factorises one pattern per handle, so the instances are bucketed by pattern (osqp_groups_bucket, C): one OSQPBatch per distinct
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


def bucket_by_pattern(Ps, As):
    """osqp_groups_bucket: group index of every problem (CscPattern lists), buckets numbered by first appearance."""
    count = len(Ps)
    pp = (C.c_void_p * count)(*[C.cast(C.pointer(p.s), C.c_void_p) for p in Ps])
    aa = (C.c_void_p * count)(*[C.cast(C.pointer(a.s), C.c_void_p) for a in As])
    group = np.zeros(count, np.int64)
    nb = int(_lib.lib().osqp_groups_bucket(count, pp, aa, group.ctypes.data_as(_lib.IP)))
    if nb < 0:
        raise RuntimeError("osqp_groups_bucket failed")
    return group, nb


class OSQPBatchGroups:
    def __init__(self, problems, device="cuda:0", one_launch=True, **settings):
        """problems: sequence of (P, q, A, l, u) with scipy sparse P (upper triangular part is used) and A, numpy q, l, u.
        one_launch=False keeps every group on its own stream even when the set qualifies for the single launch chain."""
        import torch
        from scipy import sparse
        self.count = len(problems)
        canon = []
        for i, (P, q, A, l, u) in enumerate(problems):
            Pu = sparse.triu(sparse.csc_matrix(P), format="csc"); Pu.sort_indices()
            Ac = sparse.csc_matrix(A); Ac.sort_indices()
            canon.append((Pu, Ac))
        self.n, self.m = canon[0][0].shape[0], canon[0][1].shape[0]
        if any(c[0].shape[0] != self.n or c[1].shape[0] != self.m for c in canon):
            raise ValueError("all instances must have the same (n, m)")
        pats = [(CscPattern(Pu), CscPattern(Ac)) for Pu, Ac in canon]
        group, nb = bucket_by_pattern([p for p, _ in pats], [a for _, a in pats])      # the bucketing itself is C (osqp_groups_bucket)
        buckets = {g: [] for g in range(nb)}
        for i, g in enumerate(group):
            buckets[int(g)].append(i)
        dev = torch.device(device)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        self.groups = []
        pool = [torch.cuda.Stream(device=dev) for _ in range(min(8, len(buckets)))]   # (a stream per group only pays for few groups)
        for gi, idx in enumerate(buckets.values()):
            Pu, Ac = canon[idx[0]]
            Px = np.stack([canon[i][0].data for i in idx]); Ax = np.stack([canon[i][1].data for i in idx])
            q = np.stack([np.asarray(problems[i][1], float) for i in idx])
            l = np.stack([np.asarray(problems[i][3], float) for i in idx]); u = np.stack([np.asarray(problems[i][4], float) for i in idx])
            stream = pool[gi % len(pool)]
            torch.cuda.synchronize(dev)                              # uploads above ran on torch's current stream
            w = OSQPBatch(pats[idx[0]][0], pats[idx[0]][1], t(Px), t(Ax), t(q), t(l), t(u), stream=stream, **settings)
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
        # the groups whose patterns run on the tile kernels form ONE set (a handful of launches for all of them); a group whose pattern
        # is off those kernels (a few per hundred random patterns) keeps its own stream and is solved beside the set
        L = _lib.lib()
        self._in_set = [int(L.osqp_batch_multi_key(w.h)) >= 0 for _, w in self.groups] if one_launch else [False] * len(self.groups)
        members = [k for k, ok in enumerate(self._in_set) if ok]
        if members:
            hs = (C.c_void_p * len(members))(*[self.groups[k][1].h for k in members])
            rows = torch.cat([self.groups[k][0] for k in members])            # caller row of every stacked instance of the set
            self._set_rows = rows
            whole = len(members) == len(self.groups)
            # every group in the set: the set writes straight into caller order; else into compact arrays of its own that solve() spreads
            dest = np.ascontiguousarray(rows.cpu().numpy(), dtype=np.int64) if whole else np.arange(int(rows.numel()), dtype=np.int64)
            mh = C.c_void_p()
            rc = L.osqp_multi_create(C.byref(mh), hs, len(members), dest.ctypes.data_as(_lib.IP), C.c_void_p(self._mstream.cuda_stream))
            if rc == 0:
                self._multi = mh
                self._members = members
                Bs, n, m = int(rows.numel()), self.n, self.m
                f = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
                i = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)
                self._out = dict(x=f(Bs, n), y=f(Bs, m), z=f(Bs, m), status=i(Bs), iter=i(Bs), obj=f(Bs), pri_res=f(Bs), dua_res=f(Bs))
                self._whole = whole
            elif rc != 2:
                raise RuntimeError("osqp_multi_create failed (%d)" % rc)
        if self._multi is None:
            self._in_set = [False] * len(self.groups)

    @property
    def n_patterns(self):
        return len(self.groups)

    @property
    def one_launch(self):
        """True when a solve of (nearly) all groups is one launch chain (osqp_multi_solve); groups_outside_the_chain counts the rest."""
        return self._multi is not None

    @property
    def groups_outside_the_chain(self):
        return sum(1 for ok in self._in_set if not ok)

    def solve(self):
        """Solve every group and return results in the original instance order (the arrays are reused by the next solve)."""
        import torch
        rest = [(idx, w) for (idx, w), ok in zip(self.groups, self._in_set) if not ok]
        for _, w in rest:                                    # the groups outside the set run on their own streams beside it
            w.solve_async()
        sub = None
        if self._multi is not None:
            L = _lib.lib()
            rc = L.osqp_multi_solve(self._multi)
            if rc == 2:                                      # a member's settings left the fixed-iteration case: per-workspace route from now on
                L.osqp_multi_free(self._multi)
                self._multi = None
                self._in_set = [False] * len(self.groups)
                for _, w in rest:
                    w.wait(clone=False)
                return self.solve()
            if rc:
                raise RuntimeError("osqp_multi_solve failed (%d)" % rc)
            o = self._out
            p = lambda t: C.c_void_p(t.data_ptr())
            if L.osqp_multi_get(self._multi, p(o["x"]), p(o["y"]), p(o["z"]), p(o["status"]), p(o["iter"]), p(o["obj"]), p(o["pri_res"]), p(o["dua_res"])):
                raise RuntimeError("osqp_multi_get failed")
            if self._whole and not rest:
                return dict(o)
            sub = o
        parts, err = [], None
        for _, w in rest:                                    # wait for every group before raising: a verdict must not linger in another group
            try:
                parts.append(w.wait(clone=False))
            except RuntimeError as e:
                err = e
        if err is not None:
            raise err
        if sub is None:
            # one concatenation + one gather per result field (not one indexed copy per field and group)
            return {key: torch.cat([res[key] for res in parts], 0)[self._inv] for key in parts[0]}
        out = {}
        rows_rest = torch.cat([idx for idx, _ in rest]) if rest else None
        for key, val in sub.items():
            full = torch.empty((self.count,) + tuple(val.shape[1:]), dtype=val.dtype, device=val.device)
            full[self._set_rows] = val
            if rest:
                full[rows_rest] = torch.cat([res[key] for res in parts], 0)
            out[key] = full
        return out

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
        rest = [k for k, ok in enumerate(self._in_set) if not ok]
        if self._multi is not None:
            self._mstream.wait_event(ready)
            mv = [values[k] for k in self._members]
            G = len(mv)
            px = (C.c_void_p * G)(*[v[0].data_ptr() for v in mv])
            ax = (C.c_void_p * G)(*[v[1].data_ptr() for v in mv])
            rc = _lib.lib().osqp_multi_update_P_A(self._multi, px, ax)
            # the chain reads the caller's arrays on another stream: instead of telling the caching allocator about every one of
            # them (record_stream: a millisecond of Python for a thousand groups), the producer stream waits for the chain's
            # reads -- whatever the caller does next with these arrays, or with memory the allocator hands out again on that
            # stream, runs behind them
            done = torch.cuda.Event()
            done.record(self._mstream)
            torch.cuda.current_stream(self._dev).wait_event(done)
            if rc == 2:
                rest = list(range(len(self.groups)))
            elif rc:
                raise RuntimeError("osqp_multi_update_P_A failed (%d)" % rc)
        for k in rest:
            (_, w), (Px, Ax) = self.groups[k], values[k]
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

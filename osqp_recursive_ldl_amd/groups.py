"""Batches whose instances do NOT share one sparsity pattern ("per-instance pattern" variant of config 2).

The backend factorises one pattern per handle, so the instances are bucketed by pattern: one OSQPBatch per distinct
(P pattern, A pattern), each on its own HIP stream, all enqueued before any is waited for -- with a fixed number of
iterations nothing in a solve needs the host, so the groups share the GPU concurrently -- and the results are scattered
back into the caller's instance order.  All instances must have the same (n, m).
"""
import numpy as np

from .linsys import CscPattern
from .osqp_batch import OSQPBatch


def _key(P, A):
    return (P.shape, A.shape, P.indptr.tobytes(), P.indices.tobytes(), A.indptr.tobytes(), A.indices.tobytes())


class OSQPBatchGroups:
    def __init__(self, problems, device="cuda:0", **settings):
        """problems: sequence of (P, q, A, l, u) with scipy sparse P (upper triangular part is used) and A, numpy q, l, u."""
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

    @property
    def n_patterns(self):
        return len(self.groups)

    def solve(self):
        """Solve every group (concurrently where the settings allow) and return results in the original instance order."""
        import torch
        for _, w in self.groups:
            w.solve_async()
        parts = [w.wait(clone=False) for _, w in self.groups]
        # one concatenation + one gather per result field (not one indexed copy per field and group)
        return {key: torch.cat([res[key] for res in parts], 0)[self._inv] for key in parts[0]}

    def cleanup(self):
        for _, w in self.groups:
            w.cleanup()
        self.groups = []

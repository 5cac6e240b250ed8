"""Multi-GPU layer: the batch shards embarrassingly (every QP instance is a closed system).

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU for tests).
The data path has NO collective; the only exchange is one all-gather of the packed per-instance result
record  [x(n) | y(m) | obj | pri_res | dua_res | iter | status]  at the end of a solve.
"""


def shard_range(batch, rank, world):
    """Contiguous shard [lo, hi) of `batch` instances for `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(batch), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_results(res, n, m):
    """Pack a results dict (tensors [b, ...]) into one float64 record tensor [b, n+m+5]."""
    import torch
    b = res["x"].shape[0]
    rec = torch.empty((b, n + m + 5), dtype=torch.float64, device=res["x"].device)
    rec[:, :n] = res["x"]
    rec[:, n:n + m] = res["y"]
    rec[:, n + m] = res["obj"]
    rec[:, n + m + 1] = res["pri_res"]
    rec[:, n + m + 2] = res["dua_res"]
    rec[:, n + m + 3] = res["iter"].to(torch.float64)
    rec[:, n + m + 4] = res["status"].to(torch.float64)
    return rec


def unpack_results(rec, n, m):
    import torch
    return dict(x=rec[:, :n], y=rec[:, n:n + m], obj=rec[:, n + m], pri_res=rec[:, n + m + 1], dua_res=rec[:, n + m + 2],
                iter=rec[:, n + m + 3].to(torch.int32), status=rec[:, n + m + 4].to(torch.int32))


def gather_results(res, n, m, sizes=None, force=False):
    """All-gather the per-rank result shards into the full batch on every rank (the one collective of
    the path).  `sizes` = per-rank shard sizes when they differ; equal shards use all_gather_into_tensor.
    force=True issues the collective even in a world of one rank (bench.py --force-dist: the RCCL call on a single GPU)."""
    import torch
    import torch.distributed as dist
    rec = pack_results(res, n, m).contiguous()
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return unpack_results(rec, n, m)
    world = dist.get_world_size()
    if sizes is None:
        sizes = [rec.shape[0]] * world
    smax = max(sizes)
    if rec.shape[0] < smax:                              # ragged shards: pad to the largest, trim after the gather
        pad = torch.zeros((smax - rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
        rec = torch.cat([rec, pad], 0)
    if dist.get_backend() != "gloo":
        out = torch.empty((world * smax, rec.shape[1]), dtype=rec.dtype, device=rec.device)
        dist.all_gather_into_tensor(out, rec)          # RCCL: one collective over xGMI
        parts = list(out.view(world, smax, rec.shape[1]).unbind(0))
    else:
        parts = [torch.empty((smax, rec.shape[1]), dtype=rec.dtype, device=rec.device) for _ in range(world)]
        dist.all_gather(parts, rec)
    out = torch.cat([p[:s] for p, s in zip(parts, sizes)], 0)
    return unpack_results(out, n, m)


class ResultGather:
    """The path's collective, pipelined: step k's records are packed by ONE launch (pack_fn, e.g. OSQPBatch.pack_results) into one of
    two send buffers and all-gathered asynchronously (all_gather_into_tensor, RCCL's own stream), so step k + 1's kernels do not
    wait for the exchange; a buffer pair is reused only after its collective has completed.  Equal shards only (weak scaling)."""

    def __init__(self, pack_fn, batch, n, m, device):
        import torch
        import torch.distributed as dist
        self.n, self.m, self.pack_fn = n, m, pack_fn
        world = dist.get_world_size()
        L = n + m + 5
        self.rec = [torch.empty((batch, L), dtype=torch.float64, device=device) for _ in range(2)]
        self.out = [torch.empty((world * batch, L), dtype=torch.float64, device=device) for _ in range(2)]
        self.work = [None, None]
        self.k = 0

    def gather(self):
        """Enqueue pack + all-gather of the workspace's current results; returns the index of the buffer pair."""
        import torch.distributed as dist
        i = self.k & 1
        self.k += 1
        if self.work[i] is not None:
            self.work[i].wait()                          # (stream-level wait: the buffers of two steps ago are free again)
        self.pack_fn(self.rec[i])
        self.work[i] = dist.all_gather_into_tensor(self.out[i], self.rec[i], async_op=True)
        return i

    def finish(self):
        """Wait (stream-level) for the outstanding collectives; returns the results of the last gather as a dict of views."""
        for wk in self.work:
            if wk is not None:
                wk.wait()
        self.work = [None, None]
        return unpack_results(self.out[(self.k - 1) & 1], self.n, self.m) if self.k else None


def shard_sizes(batch, world):
    return [shard_range(batch, r, world)[1] - shard_range(batch, r, world)[0] for r in range(world)]


def sharded_values(values_fn, batch, rank, world):
    """Generate only this rank's slice of a synthetic batch: values_fn(count, seed0) -> tuple of arrays."""
    lo, hi = shard_range(batch, rank, world)
    return values_fn(hi - lo, lo)


__all__ = ["shard_range", "shard_sizes", "pack_results", "unpack_results", "gather_results", "ResultGather", "sharded_values"]

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched direct KKT backend (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch of synthetic QPs that is already resident in HBM:
numeric KKT assembly + LDL' factorisation (update_matrices) followed by 200 fused ADMM iterations
(config: n=50, m=100, density 0.15, fp64, shared sparsity pattern, batch=4096 per GPU, rho=0.1,
sigma=1e-6, alpha=1.6, adaptive_rho=0, check_termination=0, scaling=0, warm_start=0 -- SURVEY.md 8d).
The host-side symbolic analysis (once per sparsity pattern) is outside the timed region.

Multi-GPU: one process per GPU under torch.distributed.run; the batch shards with no data-path
collective (weak scaling: 4096 instances per GPU) and each step ends with the one RCCL all-gather of
the packed result records.  value = instances solved by all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line with `roofline` (fused ADMM-iteration kernel, HIP-event timed on its own
stream) and, at N=1, `cpu_baseline` (the CPU oracle = a scalar port of the reference path, timed on a
bounded sample of the same workload on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--iters", type=int, default=200, help="ADMM iterations per solve")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-steps", action="store_true", help="blocking update_P_A / solve calls (host round trips inside a step)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    settings = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=args.iters, check_termination=0, adaptive_rho=0,
                    warm_start=0, scaling=0)
    if torch.cuda.device_count() == 0:                       # (does not initialise the GPU)
        raise SystemExit("bench.py needs a GPU: the backend has no CPU fallback")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # timed BEFORE the GPU is initialised: the all-core leg forks worker processes, which must not inherit a
        # live HIP context.  Same workload, same permutation (host-side symbolic analysis, no device needed).
        import osqp_recursive_ldl_amd as R0
        wl0 = R0.workloads.SharedPatternQPs(n=50, m=100, density=0.15, pattern_seed=1000)
        perm0 = R0.symbolic_analyze(wl0.P_pattern, wl0.A_pattern)["perm"]
        cpu = cpu_baseline(wl0, settings, perm0, args.cpu_seconds)
        cpu["all_cores"] = cpu_baseline_all_cores(wl0, settings, perm0, args.cpu_seconds)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the backend has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    import osqp_recursive_ldl_amd as R
    from osqp_recursive_ldl_amd import dist as rdist

    n, m, B = 50, 100, args.batch
    wl = R.workloads.SharedPatternQPs(n=n, m=m, density=0.15, pattern_seed=1000)
    Px, Ax, q, l, u = wl.values(B, seed0=rank * B)          # this rank's shard of the global batch
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dPx, dAx, dq, dl, du = t(Px), t(Ax), t(q), t(l), t(u)
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dPx, dAx, dq, dl, du, **settings)
    assert w.status == 0, "setup failed: %s" % w.status
    dims = w.linsys().dims()

    # A step is enqueued on the workspace's stream without host synchronisation (osqp_batch_update_P_A_async +
    # osqp_batch_solve_async): the GPU goes from one step's final check straight into the next step's refactorisation.
    # --sync-steps uses the blocking calls instead (one host round trip after the refactorisation, one after the solve).
    def step():
        if args.sync_steps:
            if w.update_P_A(dPx, dAx):                       # KKT value scatter + numeric factor of every instance
                raise RuntimeError("refactor failed")
            res = w.solve(clone=False)                       # 200 fused ADMM iterations + final info (views, no copies)
        else:
            if w.update_P_A(dPx, dAx, wait=False):
                raise RuntimeError("refactor could not be enqueued")
            w.solve_async()
            res = w.results(clone=False)                     # views of the workspace's result arrays (stream-ordered)
        if world > 1:
            res = rdist.gather_results(res, n, m)            # the path's only collective
        return res

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    w.wait(clone=False)
    sync()
    t0 = time.perf_counter()
    loops = []
    for _ in range(args.steps):
        res = step()
        if args.sync_steps:
            loops.append(w.last_loop())                      # HIP events around the step's ADMM loop: (ms, iterations, launches)
    if not args.sync_steps:
        w.wait(clone=False)                                  # raises if any refactorisation of the timed steps failed
        loops.append(w.last_loop())                          # (the event pair of the last step's loop)
    sync()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())

    total_instances = B * world * args.steps
    value = total_instances / elapsed
    N = n + m
    # algorithmic bytes per instance of one fused ADMM iteration: tri-solve + vector state
    tri_bytes = 8 * (dims["nnzL"] + 3 * N + m)
    iter_bytes = tri_bytes + 8 * (3 * n + 8 * m)
    loop_ms = float(np.mean([l[0] for l in loops]))
    n_iters, n_launches = loops[-1][1], loops[-1][2]
    k_ms = loop_ms / n_launches                              # average duration of one launch of the dominant kernel
    iters_per_launch = n_iters / n_launches
    achieved = iter_bytes * B * iters_per_launch / (k_ms * 1e-3) / 1e9
    out = {
        "metric": "QP solves/sec (batch) + ADMM iters/sec, n=50 m=100 fp64 batch=4096",
        "value": value, "unit": "QP solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "admm_iters_per_sec": value * args.iters,
        "config": {"workload": "random sparse QPs n=50 m=100 density=0.15, shared pattern, batch=%d per GPU, "
                               "numeric factor + %d ADMM iterations per step (rho=0.1 sigma=1e-6 alpha=1.6, "
                               "adaptive_rho=0, check_termination=0, scaling=0)" % (B, args.iters),
                   "batch_per_gpu": B, "n": n, "m": m, "nnzKKT": dims["nnzKKT"], "nnzL": dims["nnzL"],
                   "admm_iters": args.iters, "parallelism": "batch-sharded x%d, all-gather of results" % world},
        "roofline": {"bound": "hbm", "kernel": "fused ADMM iterations: rhs + permuted tri-solve + x/z/y update (k_arrow_admm on arrowhead patterns, else k_plan_admm)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "bytes_per_instance_per_iteration": iter_bytes, "tri_solve_bytes_per_instance": tri_bytes,
                     "iterations_per_launch": iters_per_launch, "kernel_ms": k_ms,
                     "note": "achieved = SURVEY 8d algorithmic bytes of one fused iteration (which assume the factor is re-streamed "
                             "every iteration) x instances x iterations per launch / launch duration; k_arrow_admm keeps the "
                             "instance's factor in registers + LDS across the iterations of a launch, so a frac above what "
                             "HBM can deliver means that re-stream is gone (see traffic)"},
    }
    # HBM traffic of the same kernel comes from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE cannot
    # be read from inside the process); it is reported only when it was measured on this very configuration
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_v14_pmc_traffic.json")))
        if B == 4096 and pmc.get("algorithmic_bytes_per_launch") == iter_bytes * B * iters_per_launch:
            out["roofline"]["traffic"] = pmc["traffic_bytes_per_launch"]
            out["roofline"]["traffic_source"] = "profiles/r1_v14_pmc_traffic.json (rocprofv3 --pmc, separate passes)"
    except (OSError, ValueError):
        pass
    status = res["status"]
    out["config"]["status_counts"] = {str(int(k)): int((status == k).sum()) for k in torch.unique(status)}

    if cpu is not None:
        assert np.array_equal(perm0, w.linsys().export_symbolic()["perm"])
        out["cpu_baseline"] = cpu
    if rank == 0:
        print(json.dumps(out))
    w.cleanup()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(wl, settings, perm, budget_s):
    """The CPU oracle (scalar C port of the reference path: form_KKT, permute, QDLDL-contract factor,
    200 ADMM iterations with one tri-solve each) on a bounded sample of the same workload, one core."""
    import ctypes as C
    import numpy as np
    import oracle_bindings as ob

    L = ob.lib()
    st = ob.settings(**settings)
    P, A = wl.P_pattern, wl.A_pattern
    Pp = np.ascontiguousarray(P.indptr, np.int64); Pi = np.ascontiguousarray(P.indices, np.int64)
    Ap = np.ascontiguousarray(A.indptr, np.int64); Ai = np.ascontiguousarray(A.indices, np.int64)
    pm = np.ascontiguousarray(perm, np.int64)

    def run(count, seed0):
        Px, Ax, q, l, u = wl.values(count, seed0=seed0)
        tf, ts = C.c_double(0), C.c_double(0)
        tot = L.orc_bench_shared_pattern(count, wl.n, wl.m, ob.ip(Pp), ob.ip(Pi), ob.fp(Px), ob.ip(Ap), ob.ip(Ai), ob.fp(Ax),
                                         ob.fp(q), ob.fp(l), ob.fp(u), C.byref(st), ob.ip(pm), None, None, C.byref(tf),
                                         C.byref(ts))
        return tot, tf.value, ts.value

    run(16, 0)                                               # warm-up (page faults, caches)
    tot, _, _ = run(64, 0)                                   # calibration
    count = int(max(64, min(100000, budget_s / max(tot / 64, 1e-6))))
    tot, tf, ts = run(count, 0)
    return {"value": count / tot, "unit": "QP solves/s", "cores": 1, "kind": "port",
            "sample": "%d instances of the bench workload (setup incl. symbolic+factor, then %d ADMM iterations), "
                      "CPU oracle built -O2, single thread" % (count, settings["max_iter"]),
            "host_cores_available": os.cpu_count(), "seconds": tot, "setup_seconds": tf, "solve_seconds": ts}


def _cpu_worker(job):
    wl, settings, perm, count, seed0 = job
    return _oracle_run(wl, settings, perm, count, seed0)


def _oracle_run(wl, settings, perm, count, seed0):
    import ctypes as C
    import numpy as np
    import oracle_bindings as ob
    L = ob.lib()
    st = ob.settings(**settings)
    P, A = wl.P_pattern, wl.A_pattern
    Pp = np.ascontiguousarray(P.indptr, np.int64); Pi = np.ascontiguousarray(P.indices, np.int64)
    Ap = np.ascontiguousarray(A.indptr, np.int64); Ai = np.ascontiguousarray(A.indices, np.int64)
    pm = np.ascontiguousarray(perm, np.int64)
    Px, Ax, q, l, u = wl.values(count, seed0=seed0)
    tf, ts = C.c_double(0), C.c_double(0)
    tot = L.orc_bench_shared_pattern(count, wl.n, wl.m, ob.ip(Pp), ob.ip(Pi), ob.fp(Px), ob.ip(Ap), ob.ip(Ai), ob.fp(Ax),
                                     ob.fp(q), ob.fp(l), ob.fp(u), C.byref(st), ob.ip(pm), None, None, C.byref(tf), C.byref(ts))
    return tot


def cpu_baseline_all_cores(wl, settings, perm, budget_s):
    """Same oracle, one instance per task over all host cores of the box's share (the reference is single-threaded
    per instance, qdldl_interface.c:208-209): forked workers, each timing only its oracle calls."""
    import multiprocessing as mp
    cores = max(1, min(16, os.cpu_count() or 1))
    per = _oracle_run(wl, settings, perm, 8, 0) / 8
    count = int(max(16, min(4000, 0.5 * budget_s / max(per, 1e-6))))
    with mp.get_context("fork").Pool(cores) as pool:
        t0 = time.perf_counter()
        times = pool.map(_cpu_worker, [(wl, settings, perm, count, 100000 + k * count) for k in range(cores)])
        wall = time.perf_counter() - t0
    return {"value": cores * count / max(times), "unit": "QP solves/s", "cores": cores,
            "sample": "%d instances per worker, %d forked workers" % (count, cores), "slowest_worker_seconds": max(times),
            "wall_seconds_incl_data_generation": wall}


if __name__ == "__main__":
    main()

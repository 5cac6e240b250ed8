#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched direct KKT backend (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch of synthetic QPs that is already resident in HBM: new values of P
and A for every instance (the nominal values times 1 + 0.05 N(0,1) per entry, a different draw every step -- config 5's
"per-step P/A perturbation", tests/update_matrices/generate_problem.py:30-33; the diagonal of P only grows, so every
instance stays convex; the eight value sets the steps cycle through are input data, formed before the timed region;
--perturb-in-step forms the products inside the step as rounds 1-2 did), numeric KKT assembly + LDL' factorisation (update_matrices) and 200 fused ADMM iterations
(config: n=50, m=100, density 0.15, fp64, shared sparsity pattern, batch=4096 per GPU, rho=0.1, sigma=1e-6, alpha=1.6,
adaptive_rho=0, check_termination=0, scaling=0, warm_start=0 -- SURVEY.md 8d).  The host-side symbolic analysis (once
per sparsity pattern) is outside the timed region.

Multi-GPU: one process per GPU.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process BEFORE anything touches torch or
HIP and relays its output; under torch.distributed.run the ranks read RANK / LOCAL_RANK / WORLD_SIZE.  The batch shards
with no data-path collective and each step ends with the one RCCL all-gather of the packed result records.  Default is
weak scaling (--batch instances per GPU); --total-batch T shards T instances over the ranks (BASELINE config 4:
--gpus 8 --total-batch 65536 = 8192 per GPU, "scaling": "strong").  value = instances solved by all ranks /
max-over-ranks time.

Rank 0 prints ONE JSON line with
  roofline        the kernel the north-star names: the batched permuted tri-solve (plugin `solve`), HBM-bound, timed live with
                  HIP events on its own stream over a rotation of handles (every launch streams from HBM) -- outside the timed
                  steps; roofline_resident = the same handle back to back (rows served by the Infinity Cache);
  roofline_fused  the kernel that dominates a step: the resident fused ADMM-iteration kernel, bound by the CU's LDS array;
  cpu_baseline    (N = 1) the CPU oracle = scalar port of the reference path, -O2 and -Ofast builds, one core and all usable
                  host cores, on a bounded sample of the same workload.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
N_CU, CLOCK_HZ = 256, 2.4e9
PERTURB_POOL = 8        # distinct per-step perturbation draws kept on the device (cycled)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed steps (default: about one second of GPU time)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU (weak scaling)")
    ap.add_argument("--total-batch", type=int, default=0, help="instances over ALL GPUs (strong scaling; config 4: 65536)")
    ap.add_argument("--iters", type=int, default=200, help="ADMM iterations per solve")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-perturb", action="store_true", help="re-upload the same P / A values every step")
    ap.add_argument("--perturb-in-step", action="store_true",
                    help="form each step's perturbed P / A values inside the timed step (two elementwise kernels per step, as rounds "
                         "1-2 were timed) instead of cycling through value sets formed before the timed region")
    ap.add_argument("--sync-steps", action="store_true", help="blocking update_P_A / solve calls (host round trips inside a step)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the result all-gather even with ONE rank (exercises the "
                         "collective path on a single GPU; the gather is then inside every timed step)")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv, script=None):
    """--gpus N > 1 without a launcher: start the N ranks as a child torch.distributed.run (this process has not touched
    torch or HIP), relay its stdout / stderr, return its exit code.  script: the program the ranks run (default: this file)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def shard_plan(args, world):
    """(instances of this job over all ranks, per-rank sizes, scaling word)"""
    from osqp_recursive_ldl_amd import dist as rdist
    if args.total_batch > 0:
        return args.total_batch, rdist.shard_sizes(args.total_batch, world), "strong"
    return args.batch * world, [args.batch] * world, "weak"


class GpuShard:
    """This rank's shard of the batch on its GPU: the workspace (osqp_batch_*), the nominal P / A values and the pool of
    per-step perturbation factors.  step() only ENQUEUES (update_P_A_async + solve_async) unless --sync-steps."""

    def __init__(self, args, wl, arrays, settings, dev, seed):
        import numpy as np
        import torch
        import osqp_recursive_ldl_amd as R
        self.args, self.torch = args, torch
        Px, Ax, q, l, u = arrays
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.dPx, self.dAx = t(Px), t(Ax)
        self.w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, self.dPx, self.dAx, t(q), t(l), t(u), **settings)
        assert self.w.status == 0, "setup failed: %s" % self.w.status
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        diag = torch.from_numpy(np.ascontiguousarray(wl.P_cols == wl.P_rows)).to(dev)
        self.fP, self.fA = [], []
        if not args.no_perturb:                              # draws made outside the timed region; the products are formed inside it
            for _ in range(PERTURB_POOL):
                e = torch.randn(self.dPx.shape, dtype=torch.float64, device=dev, generator=gen)
                self.fP.append(1.0 + 0.05 * torch.where(diag, e.abs(), e))
                self.fA.append(1.0 + 0.05 * torch.randn(self.dAx.shape, dtype=torch.float64, device=dev, generator=gen))
        self.sPx, self.sAx = torch.empty_like(self.dPx), torch.empty_like(self.dAx)
        # the PERTURB_POOL value sets a step cycles through are INPUT DATA: formed here, resident in HBM before the timed region
        # starts (8 x 32 MB at batch 4096); --perturb-in-step forms each step's products inside the step instead (two extra
        # elementwise kernels per step, 34 us at batch 4096: how rounds 1-2 were timed)
        self.pool = []
        if not args.no_perturb and not args.perturb_in_step:
            self.pool = [(self.dPx * self.fP[k], self.dAx * self.fA[k]) for k in range(PERTURB_POOL)]
        self.k = 0
        self.loops = []

    def step(self):
        w, args, torch = self.w, self.args, self.torch
        if args.no_perturb:
            px, ax = self.dPx, self.dAx
        else:
            k = self.k % PERTURB_POOL
            self.k += 1
            if self.pool:
                px, ax = self.pool[k]                            # this step's P and A values (a different set than the step before)
            else:
                torch.mul(self.dPx, self.fP[k], out=self.sPx)    # (torch's stream = the workspace's: the legacy default stream)
                torch.mul(self.dAx, self.fA[k], out=self.sAx)
                px, ax = self.sPx, self.sAx
        if args.sync_steps:
            if w.update_P_A(px, ax):                         # KKT value scatter + numeric factor of every instance
                raise RuntimeError("refactor failed")
            res = w.solve(clone=False)                       # 200 fused ADMM iterations + final info (views, no copies)
            self.loops.append(w.last_loop())                 # HIP events around the step's ADMM loop: (ms, iterations, launches)
            return res
        if w.update_P_A(px, ax, wait=False):
            raise RuntimeError("refactor could not be enqueued")
        w.solve_async()
        return w.results(clone=False)                        # views of the workspace's result arrays (stream-ordered)

    def finish(self):
        if not self.args.sync_steps:
            self.w.wait(clone=False)                         # raises if a refactorisation of the enqueued steps failed
            self.loops.append(self.w.last_loop())            # (the event pair of the last step's loop)

    def results(self):
        return self.w.results(clone=False)

    def gatherer(self, n, m):
        """The per-step collective of this shard: one pack launch + an asynchronous RCCL all-gather, double-buffered (dist.ResultGather)."""
        from osqp_recursive_ldl_amd import dist as rdist
        return rdist.ResultGather(self.w.pack_results, self.w.batch, n, m, self.dPx.device)

    def device_sync(self):
        self.torch.cuda.synchronize()


def run_steps(args, shard, n, m, sizes, scaling, world):
    """The distributed skeleton of the benchmark (shared with tests/test_bench_dist.py, which drives it under gloo with a
    CPU stand-in for `shard`): W warm-up steps, then K timed steps bracketed by barrier + device sync, MAX over ranks.
    Returns (elapsed seconds, per-rank seconds, last results of the whole batch, gather ms or None)."""
    import torch
    import torch.distributed as dist
    from osqp_recursive_ldl_amd import dist as rdist
    gsz = sizes if scaling == "strong" else None
    coll = world > 1 or getattr(args, "force_dist", False)    # --force-dist: the collective runs even in a world of one rank

    # equal shards on the GPU: packed by one launch and gathered asynchronously, so the next step does not wait for the exchange
    gat = shard.gatherer(n, m) if coll and gsz is None and hasattr(shard, "gatherer") and dist.get_backend() != "gloo" else None

    def one():
        res = shard.step()
        if gat is not None:
            gat.gather()                                     # the path's only collective
            return res
        return rdist.gather_results(res, n, m, sizes=gsz, force=coll) if coll else res

    def sync():
        if coll:
            dist.barrier()
        shard.device_sync()

    for _ in range(args.warmup):
        one()
    shard.finish()
    if gat is not None:
        gat.finish()
    sync()
    t0 = time.perf_counter()
    res = None
    for _ in range(args.steps):
        res = one()
    shard.finish()
    if gat is not None:
        res = gat.finish()                                   # the gathered results of the last step (all ranks' instances)
    sync()
    mine = time.perf_counter() - t0
    per_rank = [mine]
    if coll:
        tt = torch.tensor([mine], dtype=torch.float64, device=res["x"].device)
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        per_rank = [float(x.item()) for x in allt]
    gather_ms = None
    if coll:                                                 # the collective of one step alone, outside the timed steps
        sync()
        g0 = time.perf_counter()
        for _ in range(10):
            if gat is not None:
                gat.gather()
            else:
                rdist.gather_results(shard.results(), n, m, sizes=gsz, force=True)
        if gat is not None:
            gat.finish()
        sync()
        gather_ms = 1e2 * (time.perf_counter() - g0)
    return max(per_rank), per_rank, res, gather_ms


def check_world(args):
    """--gpus against the launcher's WORLD_SIZE; returns (world, rank, local_rank) or exits."""
    env_world = os.environ.get("WORLD_SIZE")
    world = int(env_world or "1")
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a wrong n_gpus\n" % (args.gpus, world))
        sys.exit(2)
    return world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if os.environ.get("WORLD_SIZE") is None and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))
    world, rank, local_rank = check_world(args)
    # stdout carries exactly ONE line, the JSON record of rank 0: native libraries print there too (RCCL writes its version banner
    # to stdout when the first communicator is created), so file descriptor 1 points at stderr for the whole run and the record
    # goes out through the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    settings = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=args.iters, check_termination=0, adaptive_rho=0,
                    warm_start=0, scaling=0)
    if torch.cuda.device_count() == 0:                       # (does not initialise the GPU)
        raise SystemExit("bench.py needs a GPU: the backend has no CPU fallback")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # timed BEFORE the GPU is initialised (host threads only, no fork).  Same workload, same permutation
        # (host-side symbolic analysis, no device needed).
        import osqp_recursive_ldl_amd as R0
        wl0 = R0.workloads.SharedPatternQPs(n=50, m=100, density=0.15, pattern_seed=1000)
        perm0 = R0.symbolic_analyze(wl0.P_pattern, wl0.A_pattern)["perm"]
        cpu = cpu_baseline(wl0, settings, perm0, args.cpu_seconds)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the backend has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll = world > 1 or args.force_dist
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                                       # --force-dist without a launcher: a world of this one rank
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=dev)

    import osqp_recursive_ldl_amd as R

    n, m = 50, 100
    total, sizes, scaling = shard_plan(args, world)
    lo, B = sum(sizes[:rank]), sizes[rank]
    wl = R.workloads.SharedPatternQPs(n=n, m=m, density=0.15, pattern_seed=1000)
    shard = GpuShard(args, wl, wl.values(B, seed0=lo), settings, dev, 12345 + rank)   # this rank's contiguous shard of the global batch
    w = shard.w
    dims = w.linsys().dims()

    elapsed, per_rank, res, gather_ms = run_steps(args, shard, n, m, sizes, scaling, world)
    loops = shard.loops[-args.steps:] if args.sync_steps else shard.loops[-1:]

    value = total * args.steps / elapsed
    N = n + m
    # algorithmic bytes per instance: one tri-solve; one fused ADMM iteration = tri-solve + vector state
    tri_bytes = 8 * (dims["nnzL"] + 3 * N + m)
    iter_bytes = tri_bytes + 8 * (3 * n + 8 * m)
    loop_ms = float(np.mean([lp[0] for lp in loops]))
    n_iters, n_launches = loops[-1][1], loops[-1][2]
    k_ms = loop_ms / n_launches                              # average duration of one launch of the fused kernel
    iters_per_launch = n_iters / n_launches

    # ---- roofline of the north-star kernel: the plugin `solve` (batched permuted tri-solve) on this rank's batch ----
    # Two live timings (HIP events on the kernel's stream, 200 launches each, best of three):
    #   hbm      a ROTATION of ROT handles with their own factor / tile arrays and right-hand sides (ROT x working set > the 256 MB
    #            Infinity Cache, cyclic order): every launch streams its rows from HBM -- the figure `roofline` reports;
    #   resident the same handle back to back (what an ADMM loop through the plugin API does: one factorisation, one solve per
    #            iteration): the 82 MB working set stays in the Infinity Cache between launches -- `roofline_resident`.
    ls = w.linsys()
    rhs = torch.randn((B, N), dtype=torch.float64, device=dev)
    ls.time_solve(rhs, reps=20)                              # warm-up
    solve_ms = min(ls.time_solve(rhs, reps=200) for _ in range(3))
    solve_gbs = tri_bytes * B / (solve_ms * 1e-3) / 1e9
    ROT = max(2, int(np.ceil(3 * 256e6 / (tri_bytes * B)))) if tri_bytes * B < 512e6 else 1
    hbm_ms = solve_ms
    if ROT > 1:
        rho0 = torch.full((B, m), settings["rho"], dtype=torch.float64, device=dev)
        extra = [R.BatchLinsys(wl.P_pattern, wl.A_pattern, shard.dPx * shard.fP[k % len(shard.fP)] if shard.fP else shard.dPx,
                               shard.dAx * shard.fA[k % len(shard.fA)] if shard.fA else shard.dAx, settings["sigma"], rho0)
                 for k in range(ROT - 1)]
        hs, bs = [ls] + extra, [rhs] + [torch.randn_like(rhs) for _ in extra]
        for h in hs:
            h.set_cache_policy("stream")                     # a caller that cycles through handles says so: non-temporal row loads
        R.BatchLinsys.time_solve_rotating(hs, bs, reps=4 * ROT)
        hbm_ms = min(R.BatchLinsys.time_solve_rotating(hs, bs, reps=40 * ROT) for _ in range(3))
        ls.set_cache_policy("auto")
        for h in extra:
            h.free()
    hbm_gbs = tri_bytes * B / (hbm_ms * 1e-3) / 1e9
    out = {
        "metric": "QP solves/sec (batch) + ADMM iters/sec, n=50 m=100 fp64 batch=4096",
        "value": value, "unit": "QP solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "timed_region_s": elapsed,
        "admm_iters_per_sec": value * args.iters,
        "config": {"workload": "random sparse QPs n=50 m=100 density=0.15, shared pattern, %d instances (%s), per step: P and A "
                               "of every instance perturbed by 1 + 0.05 N(0,1)%s, numeric factor, %d ADMM iterations (rho=0.1 "
                               "sigma=1e-6 alpha=1.6, adaptive_rho=0, check_termination=0, scaling=0)"
                               % (total, "%d per GPU" % sizes[0] if len(set(sizes)) == 1 else "shards %s" % sizes,
                                  " -- OFF (--no-perturb)" if args.no_perturb else
                                  (" (products formed inside the step)" if args.perturb_in_step else
                                   " (%d value sets formed before the timed region and resident in HBM, cycled: every step sees other "
                                   "values than the step before)" % PERTURB_POOL), args.iters),
                   "batch_total": total, "batch_per_gpu": sizes, "n": n, "m": m, "nnzKKT": dims["nnzKKT"], "nnzL": dims["nnzL"],
                   "admm_iters": args.iters, "parallelism": "batch-sharded x%d, one all-gather of the result records per step" % world,
                   "per_rank_seconds": per_rank, "gather_ms": gather_ms,
                   "collective": ("RCCL all_gather_into_tensor of the result records inside every step, records packed by one launch, the exchange "
                                  "asynchronous and double-buffered so the next step does not wait for it (backend %s, world %d%s)"
                                  % (dist.get_backend(), world, ", --force-dist" if args.force_dist else "")) if coll else None},
        "roofline": {"bound": "hbm", "kernel": "batched permuted tri-solve + z~ epilogue (plugin `solve`: k_tile_solve3 on arrowhead "
                                               "patterns, else k_arrow_solve / k_plan_solve), rldl_batch_time_solve_rotating",
                     "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS, "traffic": None,
                     "bytes_per_instance": tri_bytes, "instances_per_launch": B, "kernel_us": 1e3 * hbm_ms,
                     "rotation": ROT, "rotation_working_set_MB": ROT * tri_bytes * B / 1e6, "cache_policy": "stream (rldl_batch_set_cache_policy 2: non-temporal loads of the factor rows)",
                     "note": "achieved = SURVEY 8d algorithmic bytes 8 (nnzL + 3 N + m) x instances / launch duration (HIP events over "
                             "%d launches on the handles' stream); the launches cycle through %d handles with their own factor / tile / "
                             "right-hand-side arrays, %d MB together, so no launch finds its rows in the 256 MB Infinity Cache: the rows come "
                             "from HBM" % (40 * ROT, ROT, ROT * tri_bytes * B // 1000000)},
        "roofline_resident": {"bound": "hbm peak as yardstick; the rows are served by the Infinity Cache",
                              "kernel": "the same kernel, the same handle 200 times back to back (an ADMM loop through the plugin API: one "
                                        "factorisation, one solve per iteration), rldl_batch_time_solve",
                              "achieved": solve_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": solve_gbs / HBM_PEAK_GBS,
                              "kernel_us": 1e3 * solve_ms, "working_set_MB": tri_bytes * B / 1e6},
    }
    # the kernel that dominates a step: the resident fused iteration kernel.  Its roof is the CU's LDS array (x~ travels
    # between lanes through LDS; the factor stays in registers): LDS-array cycles per wave-iteration come from the committed
    # rocprofv3 PMC pass (SQ_LDS_IDX_ACTIVE), the launch duration is live.
    fused = {"bound": "lds", "kernel": "fused ADMM iterations (k_tile_admm on arrowhead patterns, else k_arrow_admm / k_plan_admm)",
             "iterations_per_launch": iters_per_launch, "kernel_ms": k_ms,
             "restream_equivalent_GBs": iter_bytes * B * iters_per_launch / (k_ms * 1e-3) / 1e9,
             "restream_note": "SURVEY 8d bytes of one fused iteration (which assume the factor is re-streamed every iteration) x "
                              "instances x iterations / launch duration: NOT a roofline figure, the factor is read once per launch"}
    pmc = load_profile("r3_pmc_fused.json")
    if pmc and B == pmc.get("batch") and iters_per_launch == pmc.get("iterations_per_launch") and pmc.get("kernel") in fused["kernel"]:
        cyc = pmc["per_wave_iteration"]["SQ_LDS_IDX_ACTIVE"]
        busy = cyc * B * iters_per_launch / (k_ms * 1e-3 * N_CU * CLOCK_HZ)
        fused.update(achieved=busy * N_CU * CLOCK_HZ / 1e9, peak=N_CU * CLOCK_HZ / 1e9, unit="G LDS-array cycles/s", frac=busy,
                     lds_cycles_per_wave_iteration=cyc, counters_source="profiles/r3_pmc_fused.json (rocprofv3 --pmc, this configuration)",
                     note="frac = LDS-array busy cycles / (256 CUs x 2.4 GHz x launch duration): a lower bound of the busy fraction "
                          "(the chip clocks below 2.4 GHz under load)")
    out["roofline_fused"] = fused
    tr = load_profile("r3_pmc_traffic_solve.json")
    if tr and B == tr.get("batch") and tr.get("algorithmic_bytes_per_launch") == tri_bytes * B and "tile_solve3" in tr.get("kernel", ""):
        out["roofline"]["traffic"] = tr["traffic_bytes_per_launch"]
        out["roofline"]["traffic_source"] = "from_file: profiles/r3_pmc_traffic_solve.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
    status = res["status"]
    out["config"]["status_counts"] = {str(int(k)): int((status == k).sum()) for k in torch.unique(status)}

    if cpu is not None:
        assert np.array_equal(perm0, w.linsys().export_symbolic()["perm"])
        out["cpu_baseline"] = cpu
    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)
    w.cleanup()
    if coll:
        dist.destroy_process_group()


def load_profile(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return None


def usable_cores():
    """Host cores this process may use: the affinity mask, capped by a cgroup CPU quota when one is set."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(wl, settings, perm, budget_s):
    """The CPU oracle (scalar C port of the reference path: form_KKT, permute, QDLDL-contract factor, 200 ADMM iterations
    with one tri-solve each) on a bounded sample of the same workload: built -O2 (parity build) and -Ofast (the flags the
    reference ships with, CMakeLists.txt:152), one core and all usable host cores (one instance per task: the reference is
    single-threaded per instance, qdldl_interface.c:208-209).  Host threads inside the oracle library, no fork."""
    import ctypes as C
    import numpy as np
    import oracle_bindings as ob

    st = ob.settings(**settings)
    P, A = wl.P_pattern, wl.A_pattern
    Pp = np.ascontiguousarray(P.indptr, np.int64); Pi = np.ascontiguousarray(P.indices, np.int64)
    Ap = np.ascontiguousarray(A.indptr, np.int64); Ai = np.ascontiguousarray(A.indices, np.int64)
    pm = np.ascontiguousarray(perm, np.int64)
    ndata = 512
    Px, Ax, q, l, u = wl.values(ndata, seed0=0)
    cores = usable_cores()

    def run(L, threads, per_thread):
        return L.orc_bench_shared_pattern_mt(threads, per_thread, ndata, wl.n, wl.m, ob.ip(Pp), ob.ip(Pi), ob.fp(Px), ob.ip(Ap),
                                             ob.ip(Ai), ob.fp(Ax), ob.fp(q), ob.fp(l), ob.fp(u), C.byref(st), ob.ip(pm))

    def leg(L, threads, seconds):
        run(L, threads, 4)                                   # warm-up (page faults, caches, thread start)
        per = run(L, threads, 16) / 16                       # calibration: seconds per instance and thread
        count = int(max(16, min(200000, seconds / max(per, 1e-6))))
        wall = run(L, threads, count)
        assert wall > 0
        return {"value": threads * count / wall, "cores": threads, "instances": threads * count, "seconds": wall}

    share = budget_s / 4.0
    res = {}
    for name, L in (("O2", ob.lib()), ("Ofast", ob.lib_ofast())):
        res[name] = {"one_core": leg(L, 1, share), "all_cores": leg(L, cores, share)}
    best = res["Ofast"]["one_core"]
    return {"value": best["value"], "unit": "QP solves/s", "cores": 1, "kind": "port",
            "sample": "%d instances of the bench workload cycled from %d distinct ones (per instance: setup incl. symbolic analysis and "
                      "numeric factor, then %d ADMM iterations), CPU oracle built -Ofast (the reference's shipped flags), one thread"
                      % (best["instances"], ndata, settings["max_iter"]),
            "builds": res, "all_cores": dict(res["Ofast"]["all_cores"], unit="QP solves/s", build="-Ofast"),
            "host_cores_available": os.cpu_count(), "host_cores_usable": cores}


if __name__ == "__main__":
    main()

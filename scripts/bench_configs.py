#!/usr/bin/env python3
"""Secondary measurements for the BASELINE.json configs that are not the headline bench line
(run on the GPU box; prints one JSON object per measurement):

  solve_kernel   standalone batched `solve` (the plugin call) at batch 4096, metric shape, vs its roofline
  config2        batch 1024 metric shape: numeric factor + 200 ADMM iterations
  config3        MPC stage blocks N=20 nx=12 nu=4 ny=10 nt=12, batch 4096: full stage factorisation,
                 refactor on a rho change, restart from stage k, tri-solve
  config5        update_matrices on the metric shape (full numeric refactor) and, on the MPC shape,
                 restart-from-first-modified-stage vs full refactor
  horizon        horizon change 19 <-> 20 on the MPC shape (osqp_update_recursive): first visit, cached, with / without
                 adopting the shared factor columns, against setting the problem up from scratch
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import osqp_recursive_ldl_amd as R

PEAK = 8000.0
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def timed(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def emit(**kw):
    print(json.dumps(kw), flush=True)


def metric_shape():
    wl = R.workloads.SharedPatternQPs()
    B = 4096
    Px, Ax, q, l, u = wl.values(B)
    dPx, dAx = t(Px), t(Ax)
    rho = t(np.full((B, wl.m), 0.1))
    ls = R.BatchLinsys(wl.P_pattern, wl.A_pattern, dPx, dAx, 1e-6, rho)
    d = ls.dims()
    N = d["n"] + d["m"]
    b = torch.randn((B, N), dtype=torch.float64, device=dev)
    ms = ls.time_solve(b, reps=200)
    by = 8 * (d["nnzL"] + 3 * N + d["m"])
    emit(name="solve_kernel", batch=B, n=d["n"], m=d["m"], nnzL=d["nnzL"], us_per_launch=1e3 * ms, bytes_per_instance=by,
         achieved_GBs=by * B / (ms * 1e-3) / 1e9, frac=by * B / (ms * 1e-3) / 1e9 / PEAK)
    ms = timed(lambda: ls.update_matrices(dPx, dAx))
    fb = 8 * (d["nnzKKT"] + d["nnzL"]) + 16 * N
    emit(name="config5_update_matrices_full_refactor", batch=B, ms=ms, bytes_per_instance=fb,
         achieved_GBs=fb * B / (ms * 1e-3) / 1e9, refactors_per_sec=B / (ms * 1e-3))
    ms = timed(lambda: ls.update_rho_vec(rho))
    emit(name="update_rho_vec_full_refactor", batch=B, ms=ms, refactors_per_sec=B / (ms * 1e-3))
    ls.free()
    # config 2: batch 1024
    B2 = 1024
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=200, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dPx[:B2].contiguous(), dAx[:B2].contiguous(), t(q[:B2]), t(l[:B2]), t(u[:B2]), **kw)

    def step():
        w.update_P_A(dPx[:B2].contiguous(), dAx[:B2].contiguous())
        w.solve()
    ms = timed(step, reps=5)
    lm, its, gr = w.last_loop()
    emit(name="config2_batch1024_factor_plus_200_iters", batch=B2, ms_per_step=ms, qp_solves_per_sec=B2 / (ms * 1e-3),
         us_per_iteration=1e3 * lm / its, launches_per_solve=gr)
    w.cleanup()
    # end-to-end with the reference's default equilibration (scaling=10 for the end-to-end number)
    kw10 = dict(kw, scaling=10)
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dPx, dAx, t(q), t(l), t(u), **kw10)

    def step10():
        w.update_P_A(dPx, dAx)
        w.solve()
    ms = timed(step10, reps=5)
    emit(name="metric_shape_batch4096_scaling10_factor_plus_200_iters", batch=B, ms_per_step=ms, qp_solves_per_sec=B / (ms * 1e-3))
    # default OSQP behaviour: termination checks every 25 iterations, adaptive rho, eps 1e-3
    kwd = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=4000, check_termination=25, adaptive_rho=1, adaptive_rho_interval=100,
               eps_abs=1e-3, eps_rel=1e-3, warm_start=0, scaling=10)
    w.cleanup()
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dPx, dAx, t(q), t(l), t(u), **kwd)
    r = None

    def stepd():
        nonlocal r
        r = w.solve()
    ms = timed(stepd, reps=5)
    emit(name="metric_shape_batch4096_default_termination_scaling10", batch=B, ms_per_solve=ms, qp_solves_per_sec=B / (ms * 1e-3),
         mean_iterations=float(r["iter"].double().mean()), solved=int((r["status"] == 1).sum()))
    w.cleanup()
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dPx, dAx, t(q), t(l), t(u), polish=1, **kwd)
    ms = timed(stepd, reps=5)
    emit(name="metric_shape_batch4096_default_termination_scaling10_polish", batch=B, ms_per_solve=ms, qp_solves_per_sec=B / (ms * 1e-3),
         polished=int((r["status_polish"] == 1).sum()), max_pri_res=float(r["pri_res"].max()), max_dua_res=float(r["dua_res"].max()))
    w.cleanup()


def pattern_groups():
    """Per-instance-pattern variant of config 2: 8 different random patterns x 512 instances each."""
    probs = []
    for s in range(8):
        wl = R.workloads.SharedPatternQPs(pattern_seed=2000 + s)
        probs += [wl.instance(b) for b in range(512)]
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=200, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
    g = R.OSQPBatchGroups(probs, **kw)
    ms = timed(lambda: g.solve(), reps=3)
    emit(name="config2_pattern_groups_8x512_200_iters", batch=len(probs), patterns=g.n_patterns, one_launch_chain=bool(g.one_launch),
         ms_per_solve=ms, qp_solves_per_sec=len(probs) / (ms * 1e-3))
    g.cleanup()
    g = R.OSQPBatchGroups(probs, one_launch=False, **kw)            # every group on its own stream (what patterns off the tile kernels get)
    ms = timed(lambda: g.solve(), reps=3)
    emit(name="config2_pattern_groups_8x512_200_iters_one_stream_per_pattern", batch=len(probs), patterns=g.n_patterns, ms_per_solve=ms,
         qp_solves_per_sec=len(probs) / (ms * 1e-3))
    g.cleanup()
    wl = R.workloads.SharedPatternQPs(pattern_seed=2000)           # the same solve (no refactorisation) on ONE pattern, for the ratio
    Px, Ax, q, l, u = wl.values(len(probs))
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, t(Px), t(Ax), t(q), t(l), t(u), **kw)
    ms1 = timed(lambda: w.solve(), reps=3)
    emit(name="config2_single_pattern_4096_200_iters_solve_only", batch=len(probs), ms_per_solve=ms1)
    w.cleanup()


def mpc_shape():
    wl = R.workloads.MPCStageQPs(N=20)
    B = 4096
    Px, Ax, q, l, u = wl.values(B)
    dPx, dAx = t(Px), t(Ax)
    rho = t(np.full((B, wl.m), 0.1))
    ls = R.BatchLinsys.recursive(wl.dims, wl.P_pattern, wl.A_pattern, dPx, dAx, 1e-6, rho)
    assert ls.status == 0
    d = ls.dims()
    N = d["n"] + d["m"]
    emit(name="config3_dims", batch=B, n=d["n"], m=d["m"], nnzKKT=d["nnzKKT"], nnzL=d["nnzL"])
    full = timed(lambda: ls.update_from_stage(0, dPx, dAx, None), reps=3)
    emit(name="config3_full_stage_factorisation", batch=B, ms=full, factorisations_per_sec=B / (full * 1e-3))
    rho2 = t(np.full((B, wl.m), 0.4))
    ms = timed(lambda: ls.update_from_stage(0, None, None, rho2), reps=3)
    emit(name="config3_refactor_on_rho_change", batch=B, ms=ms, refactors_per_sec=B / (ms * 1e-3))
    for k in (5, 10, 15, 19):
        ms = timed(lambda: ls.update_from_stage(k, dPx, dAx, None), reps=3)
        emit(name="config5_restart_from_stage", stage=k, batch=B, ms=ms, speedup_vs_full=full / ms)
    b = torch.randn((B, N), dtype=torch.float64, device=dev)
    ms = ls.time_solve(b, reps=50)
    by = 8 * (d["nnzL"] + 3 * N + d["m"])
    emit(name="config3_solve_kernel", batch=B, us_per_launch=1e3 * ms, bytes_per_instance=by,
         achieved_GBs=by * B / (ms * 1e-3) / 1e9, frac=by * B / (ms * 1e-3) / 1e9 / PEAK)
    ls.free()
    # ADMM on the MPC shape: stage-block factorisation + 200 fused iterations (grouped-plan kernels, factor from global memory)
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=200, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
    w = R.OSQPBatch.recursive(wl.dims, wl.Q0, wl.Qi, wl.QN, wl.A0, wl.Ai, wl.Aij, wl.AN, t(q), t(l), t(u), **kw)

    def step():
        w.update_recursive(0, dPx, dAx)
        w.solve(clone=False)
    ms = timed(step, reps=3)
    lm, its, gr = w.last_loop()
    emit(name="config3_admm_factor_plus_200_iters", batch=B, ms_per_step=ms, qp_solves_per_sec=B / (ms * 1e-3),
         us_per_iteration=1e3 * lm / its)
    w.cleanup()


def horizon_change():
    import time
    B, Nmax = 4096, 20
    wl = {N: R.workloads.MPCStageQPs(N=N) for N in (19, 20)}
    data = {}
    for N in (19, 20):
        _, _, q, l, u = wl[N].values(8)
        data[N] = [t(np.tile(a, (B // 8, 1))) for a in (q, l, u)]
    Px, Ax = wl[19].values(8)[:2]
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=200, check_termination=0, adaptive_rho=0, warm_start=1, scaling=0)
    blocks = lambda w: (w.Q0, w.Qi, w.QN, w.A0, w.Ai, w.Aij, w.AN)
    t0 = time.perf_counter()
    fresh = R.OSQPBatch.recursive(wl[20].dims, *blocks(wl[20]), *data[20], **kw)
    torch.cuda.synchronize()
    emit(name="horizon_setup_from_scratch_N20", batch=B, wall_ms=1e3 * (time.perf_counter() - t0))
    fresh.cleanup()
    for store in ("single", "multi"):                            # one workspace at Nmax dimensions / a workspace per visited horizon
        os.environ.pop("RLDL_HORIZON_MULTI", None)
        if store == "multi":
            os.environ["RLDL_HORIZON_MULTI"] = "1"
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        hz = R.OSQPHorizon(wl[19].dims, Nmax, *blocks(wl[19]), *data[19], **kw)
        hz.workspace.update_P_A(t(np.tile(Px, (B // 8, 1))), t(np.tile(Ax, (B // 8, 1))))
        hz.workspace.solve(clone=False)

        def change(N):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = hz.update(N, *data[N])
            torch.cuda.synchronize()
            assert rc == 0
            return 1e3 * (time.perf_counter() - t0)
        first = change(20)
        emit(name="horizon_change_first_visit", store=store, single_store=hz.single_store, frm=19, to=20, batch=B, wall_ms=first, **hz.last_update())
        for mode in ("adopt", "full") if store == "multi" else ("restart",):
            if mode == "full":
                os.environ["RLDL_HORIZON_FULL"] = "1"
            ts = {19: [], 20: []}
            for _ in range(5):
                ts[19].append(change(19)); info19 = hz.last_update()
                ts[20].append(change(20)); info20 = hz.last_update()
            torch.cuda.synchronize()
            emit(name="horizon_change_cached", store=store, mode=mode, batch=B, wall_ms_20_to_19=min(ts[19]), wall_ms_19_to_20=min(ts[20]),
                 reused_19=info19["instances_reused"], reused_20=info20["instances_reused"], workspaces=hz.n_workspaces,
                 device_MB_held=(free0 - torch.cuda.mem_get_info()[0]) / 1e6)
        os.environ.pop("RLDL_HORIZON_FULL", None)
        if store == "single":                                    # a far move on the same store: 20 -> 5 -> 20 (cost follows the live blocks)
            _, _, q5, l5, u5 = R.workloads.MPCStageQPs(N=5).values(8)
            d5 = [t(np.tile(a, (B // 8, 1))) for a in (q5, l5, u5)]
            torch.cuda.synchronize(); t0 = time.perf_counter(); assert hz.update(5, *d5) == 0; torch.cuda.synchronize()
            down = 1e3 * (time.perf_counter() - t0)
            ms5 = timed(lambda: hz.workspace.solve(clone=False), reps=3)
            torch.cuda.synchronize(); t0 = time.perf_counter(); assert hz.update(20, *data[20]) == 0; torch.cuda.synchronize()
            up = 1e3 * (time.perf_counter() - t0)
            ms20 = timed(lambda: hz.workspace.solve(clone=False), reps=3)
            emit(name="horizon_single_store_far_move", batch=B, wall_ms_20_to_5=down, wall_ms_5_to_20=up, solve_ms_200_iterations_N5=ms5,
                 solve_ms_200_iterations_N20=ms20, workspaces=hz.n_workspaces)
        hz.free()
    os.environ.pop("RLDL_HORIZON_MULTI", None)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "metric"):
        metric_shape()
    if which in ("all", "groups"):
        pattern_groups()
    if which in ("all", "mpc"):
        mpc_shape()
    if which in ("all", "horizon"):
        horizon_change()

"""GPU parity tests on the MPC stage-block shape (BASELINE config 3 family): the stage-interleaved permutation is
handed to the batched ADMM driver and the iterates are compared with the CPU oracle run with the same
permutation.  Exercises the grouped plan with many small dense groups, the large-N fused iteration kernel and
(for N=20) the variant that reads the factor from global memory instead of staging it in LDS.

The recursive path has no reference fixture ("parity unpinned"); the oracle here is the generic sparse LDL of the
same permuted KKT matrix, which is what the stage recursion computes."""
import numpy as np
import pytest

import oracle_bindings as ob

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to("cuda:0")


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1.0, float(np.max(np.abs(b)))))


@pytest.mark.parametrize("N", [4, 20])
def test_mpc_admm_iterates_match_oracle(N):
    import osqp_recursive_ldl_amd as R
    wl = R.workloads.MPCStageQPs(N=N)
    B = 3
    Px, Ax, q, l, u = wl.values(B)
    perm = R.workloads.stage_permutation(*wl.dims)
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=60, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), perm=perm, **kw)
    assert w.status == 0
    r = w.solve()
    for b in range(B):
        P, qq, A, ll, uu = wl.instance(b)
        ro = ob.OracleOSQP(P, qq, A, ll, uu, perm=perm, **kw).solve()
        assert relerr(r["x"][b].cpu().numpy(), ro["x_iter"]) < 1e-8
        assert relerr(r["y"][b].cpu().numpy(), ro["y_iter"]) < 1e-8
        assert relerr(r["z"][b].cpu().numpy(), ro["z_iter"]) < 1e-8
        assert int(r["iter"][b]) == ro["iter"] == 60
    w.cleanup()


def test_mpc_equality_rows_get_the_stiff_rho_and_converge():
    """Dynamics rows are equalities (l == u) -> rho_vec = 1e3 * rho on them (auxil.c:88-91); the solve must reach
    OSQP_SOLVED with the same iteration count as the oracle."""
    import osqp_recursive_ldl_amd as R
    wl = R.workloads.MPCStageQPs(N=5)
    B = 2
    Px, Ax, q, l, u = wl.values(B)
    perm = R.workloads.stage_permutation(*wl.dims)
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=4000, check_termination=25, adaptive_rho=1, adaptive_rho_interval=100,
              eps_abs=1e-4, eps_rel=1e-4, warm_start=0, scaling=0)
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), perm=perm, **kw)
    r = w.solve()
    for b in range(B):
        P, qq, A, ll, uu = wl.instance(b)
        ro = ob.OracleOSQP(P, qq, A, ll, uu, perm=perm, **kw).solve()
        assert int(r["status"][b]) == ro["status"]
        assert int(r["iter"][b]) == ro["iter"]
        assert relerr(r["x"][b].cpu().numpy(), ro["x"]) < 1e-6
    w.cleanup()


def test_setup_recursive_from_stage_blocks_equals_assembled_setup():
    """osqp_setup_recursive mirror: building the workspace from the seven stage blocks gives the same iterates as handing
    over the assembled P, A with the stage permutation; per-instance values then go in through update_recursive, which
    restarts the factorisation at the first modified stage, and partial_update_bounds equals a full update_bounds."""
    import osqp_recursive_ldl_amd as R
    wl = R.workloads.MPCStageQPs(N=6)
    B = 3
    Px, Ax, q, l, u = wl.values(B)
    perm = R.workloads.stage_permutation(*wl.dims)
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=50, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
    wr = R.OSQPBatch.recursive(wl.dims, wl.Q0, wl.Qi, wl.QN, wl.A0, wl.Ai, wl.Aij, wl.AN, dev(q), dev(l), dev(u), **kw)
    assert wr.status == 0
    assert np.array_equal(wr.P.i, R.CscPattern(wl.P_pattern).i) and np.array_equal(wr.A.i, R.CscPattern(wl.A_pattern).i)
    nomP = np.tile(wr.P.x, (B, 1)); nomA = np.tile(wr.A.x, (B, 1))
    wa = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(nomP), dev(nomA), dev(q), dev(l), dev(u), perm=perm, **kw)
    ra, rr = wa.solve(), wr.solve()
    # (the recursive workspace solves by stage blocks, the assembled one with the grouped plan: same math, other order)
    assert relerr(rr["x"].cpu().numpy(), ra["x"].cpu().numpy()) < 1e-9 and relerr(rr["y"].cpu().numpy(), ra["y"].cpu().numpy()) < 1e-9
    # per-instance values: only stages >= 4 differ from the nominal problem
    first = 4
    col0 = wl.nu + (first - 1) * (wl.nx + wl.nu)
    Pp, Ap = wr.P.p, wr.A.p
    Px2, Ax2 = nomP.copy(), nomA.copy()
    Px2[:, Pp[col0]:] = Px[:, Pp[col0]:]; Ax2[:, Ap[col0]:] = Ax[:, Ap[col0]:]
    assert wr.update_recursive(first, dev(Px2), dev(Ax2)) == 0
    assert wa.update_P_A(dev(Px2), dev(Ax2)) == 0                                     # full refactor
    ra, rr = wa.solve(), wr.solve()
    assert relerr(rr["x"].cpu().numpy(), ra["x"].cpu().numpy()) < 1e-9
    for b in range(B):
        assert relerr(wr.linsys().export_factor(b)["Lx"], wa.linsys().export_factor(b)["Lx"]) < 1e-11
    # partial bounds update == full bounds update
    l2, u2 = l.copy(), u.copy()
    s0, s1 = 5, 17
    l2[:, s0:s1] -= 0.25; u2[:, s0:s1] += 0.5
    assert wr.partial_update_bounds(s0, s1, dev(l2[:, s0:s1]), dev(u2[:, s0:s1])) == 0
    assert wa.update_bounds(dev(l2), dev(u2)) == 0
    ra, rr = wa.solve(), wr.solve()
    assert relerr(rr["x"].cpu().numpy(), ra["x"].cpu().numpy()) < 1e-9
    assert wr.partial_update_bounds(3, 3, dev(l2[:, :0]), dev(u2[:, :0])) == 1         # start >= stop (recursive_ldl.c:126)
    wa.cleanup(); wr.cleanup()


@pytest.mark.parametrize("N", [1, 2, 7, 20])
def test_block_tri_solve_and_fused_iteration_match_oracle(N):
    """Stage-structured handles solve by dense stage blocks (stage_tri_solve: tiles of L_bb and L(b+1, b) staged through
    LDS, row per lane): the plugin `solve` and the fused ADMM iteration built on it against the oracle's QDLDL_solve /
    ADMM on the same permuted KKT matrix.  N = 1 has no interior stage, N = 20 is the BASELINE config 3 shape."""
    import osqp_recursive_ldl_amd as R
    wl = R.workloads.MPCStageQPs(N=N)
    B = 5                                                        # not a multiple of the 4 waves per workgroup
    Px, Ax, q, l, u = wl.values(B)
    perm = R.workloads.stage_permutation(*wl.dims)
    rho = np.where(l == u, 100.0, 0.1)
    ls = R.BatchLinsys.recursive(wl.dims, wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), 1e-6, dev(rho))
    assert ls.status == 0
    rhs = np.random.default_rng(11).standard_normal((B, wl.n + wl.m))
    sol = ls.solve(dev(rhs)).cpu().numpy()
    for b in range(B):
        P, qq, A, ll, uu = wl.instance(b)
        o = ob.OracleLinsys(P, A, 1e-6, rho[b], perm=perm)
        assert relerr(sol[b], o.solve(rhs[b])) < 1e-9
    ls.free()
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=40, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
    w = R.OSQPBatch.recursive(wl.dims, wl.Q0, wl.Qi, wl.QN, wl.A0, wl.Ai, wl.Aij, wl.AN, dev(q), dev(l), dev(u), **kw)
    assert w.update_P_A(dev(Px), dev(Ax)) == 0
    r = w.solve()
    for b in range(B):
        P, qq, A, ll, uu = wl.instance(b)
        ro = ob.OracleOSQP(P, qq, A, ll, uu, perm=perm, **kw).solve()
        assert relerr(r["x"][b].cpu().numpy(), ro["x_iter"]) < 1e-8
        assert relerr(r["y"][b].cpu().numpy(), ro["y_iter"]) < 1e-8
        assert relerr(r["z"][b].cpu().numpy(), ro["z_iter"]) < 1e-8
    w.cleanup()


def test_iterations_of_one_launch_equal_one_launch_per_iteration():
    """Stage handles run a whole group of ADMM iterations in one launch (k_plan_admm_loop with the product tri-solve: an iteration's
    update leaves the next right-hand side on chip).  Thirty iterations in one solve and thirty warm-started solves of one
    iteration each are the same arithmetic on the same values: bit-identical iterates."""
    import osqp_recursive_ldl_amd as R
    wl = R.workloads.MPCStageQPs(N=5)
    B = 6
    Px, Ax, q, l, u = wl.values(B)
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
    w30 = R.OSQPBatch.recursive(wl.dims, wl.Q0, wl.Qi, wl.QN, wl.A0, wl.Ai, wl.Aij, wl.AN, dev(q), dev(l), dev(u), max_iter=30, **kw)
    w1 = R.OSQPBatch.recursive(wl.dims, wl.Q0, wl.Qi, wl.QN, wl.A0, wl.Ai, wl.Aij, wl.AN, dev(q), dev(l), dev(u), max_iter=1, **kw)
    assert w30.update_P_A(dev(Px), dev(Ax)) == 0 and w1.update_P_A(dev(Px), dev(Ax)) == 0
    r30 = w30.solve()
    r1 = w1.solve()                                                 # cold start, one iteration
    assert w1.update_settings(warm_start=1) == 0
    for _ in range(29):
        r1 = w1.solve()
    for key in ("x_iter", "y_iter", "z", "delta_x", "delta_y"):
        assert torch.equal(r30[key], r1[key]), key
    w30.cleanup(); w1.cleanup()


def test_full_size_mpc_batch_4096_properties():
    """BASELINE config 3 at full size (N = 20, KKT 772, batch 4096), stage-structured handle: size-independent properties
    instead of an oracle sweep -- linearity of the block tri-solve, residual of the permuted system on a spread sample,
    batch position does not matter, restart from a stage == full refactorisation, horizon 20 -> 19 -> 20 returns to the
    same factor bit for bit."""
    import osqp_recursive_ldl_amd as R
    from helpers import full_kkt
    wl = R.workloads.MPCStageQPs(N=20)
    B, rep = 4096, 16
    Px, Ax, q, l, u = wl.values(rep)
    tile = lambda a: np.tile(a, (B // rep, 1))
    dPx, dAx = dev(tile(Px)), dev(tile(Ax))
    rho = np.where(tile(l) == tile(u), 100.0, 0.1)
    ls = R.BatchLinsys.recursive(wl.dims, wl.P_pattern, wl.A_pattern, dPx, dAx, 1e-6, dev(rho))
    assert ls.status == 0 and (ls.factor_status() == wl.n).all()
    N = wl.n + wl.m
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    r1 = torch.randn((B, N), dtype=torch.float64, device="cuda:0", generator=g)
    r2 = torch.randn((B, N), dtype=torch.float64, device="cuda:0", generator=g)
    s1, s2, s3 = ls.solve(r1.clone()), ls.solve(r2.clone()), ls.solve((0.5 * r1 - r2).clone())
    assert float((s3[:, :wl.n] - (0.5 * s1[:, :wl.n] - s2[:, :wl.n])).abs().max()) < 1e-6 * max(1.0, float(s1.abs().max()))
    for b in range(0, B, 409):
        P, qq, A, ll, uu = wl.instance(b % rep)
        K = full_kkt(P, A, 1e-6, rho[b])
        out = s1[b].cpu().numpy(); r = r1[b].cpu().numpy()
        raw = np.concatenate([out[:wl.n], (out[wl.n:] - r[wl.n:]) * rho[b]])
        assert np.max(np.abs(K @ raw - r)) < 1e-7 * max(1.0, np.max(np.abs(raw)))
    assert bool(torch.isfinite(s3).all())
    f0 = ls.export_factor(4095)["Lx"].copy()
    assert np.array_equal(f0, ls.export_factor(4095 - rep)["Lx"])                        # same data, other batch position
    assert ls.update_from_stage(12, dPx, dAx, None) == 0                                 # restart == what was there
    import os
    if os.environ.get("RLDL_NO_STAGE_FACTOR"):       # generic kernel: the restart replays the kept columns' updates in another order
        assert relerr(ls.export_factor(4095)["Lx"], f0) < 1e-11
    else:
        assert np.array_equal(ls.export_factor(4095)["Lx"], f0)
    ls.free()
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=20, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
    w19 = R.workloads.MPCStageQPs(N=19)
    q19, l19, u19 = [dev(np.tile(a, (B // rep, 1))) for a in w19.values(rep)[2:]]
    hz = R.OSQPHorizon(wl.dims, 20, wl.Q0, wl.Qi, wl.QN, wl.A0, wl.Ai, wl.Aij, wl.AN, dev(tile(q)), dev(tile(l)), dev(tile(u)), **kw)
    assert hz.workspace.update_P_A(dPx, dAx) == 0
    ra = hz.workspace.solve()
    fa = hz.workspace.linsys().export_factor(2049)
    nre = 0 if (os.environ.get("RLDL_HORIZON_FULL") or os.environ.get("RLDL_NO_STAGE_FACTOR")) else B
    assert hz.update(19, q19, l19, u19) == 0 and hz.last_update()["instances_reused"] == nre
    r19 = hz.workspace.solve()
    assert bool(torch.isfinite(r19["x"]).all()) and (r19["iter"] == 20).all()
    assert hz.update(20, dev(tile(q)), dev(tile(l)), dev(tile(u))) == 0 and hz.last_update()["instances_reused"] == nre
    fb = hz.workspace.linsys().export_factor(2049)
    # stages >= 19 are nominal again after the round trip (update_AP_matrices), stages < 19 kept the instance's values
    sym = hz.workspace.linsys().export_symbolic()
    c19 = wl.nu + (wl.nx + wl.ny) + 18 * (2 * wl.nx + wl.nu + wl.ny)
    keep = sym["Lp"][c19]
    assert np.array_equal(fa["Lx"][:keep], fb["Lx"][:keep]) and np.array_equal(fa["D"][:c19], fb["D"][:c19])
    assert torch.equal(ra["x"][0], ra["x"][rep]) and torch.equal(r19["x"][3], r19["x"][3 + rep])
    hz.free()

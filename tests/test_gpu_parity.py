"""GPU parity tests (pytest -m gpu): the HIP path, called through the C-ABI, against the CPU oracle on
the same seeded inputs, against the committed golden fixtures, and -- at BASELINE.json's full sizes --
through size-independent properties.

Tolerances (north_star "within a stated fp64 tolerance"):
  integers (perm given, etree, Lnz, Lp, Li)            exact
  Lx, D, Dinv vs oracle                                rel 1e-12 (operation order differs: right-looking vs up-looking)
  solve output vs oracle                               rel 1e-10
  ADMM iterates after 200 fixed-rho iterations         rel 1e-8
  reference known answers (x, y, obj)                  1e-4 = TESTS_TOL (tests/minunit.h:13)
"""
import os

import numpy as np
import pytest
from scipy import sparse

import oracle_bindings as ob
from helpers import dense_from_L, full_kkt, load_golden, osqp_inf

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

TESTS_TOL = 1e-4


@pytest.fixture(scope="module")
def R():
    import osqp_recursive_ldl_amd as R
    return R


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to("cuda:0")


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1.0, float(np.max(np.abs(b)))))


# ------------------------------------------------------------------ legacy vtable (config 1 plumbing) ----
def test_legacy_vtable_solve_linsys_golden(R):
    """tests/solve_linsys/test_solve_linsys.h:12-48 through the HIP backend (same pattern the reference uses
    to test a second backend: identical suite, different solver enum)."""
    d = load_golden("solve_linsys")["data"]
    m = d["test_solve_KKT_m"]
    s = R.HipLDLSolver(d["test_solve_KKT_Pu"], d["test_solve_KKT_A"], d["test_solve_KKT_sigma"],
                       np.full(m, d["test_solve_KKT_rho"]))
    assert s.status == 0 and s.type == 20
    x = s.solve(d["test_solve_KKT_rhs"])
    assert np.max(np.abs(x - d["test_solve_KKT_x"])) < TESTS_TOL
    assert np.max(np.abs(x - d["test_solve_KKT_x"])) < 1e-10
    s.free()


def test_legacy_vtable_polish_and_updates(R):
    d = load_golden("update_matrices")["data"]
    P, A = d["test_form_KKT_Pu"], d["test_form_KKT_A"]
    n, m = d["test_form_KKT_n"], d["test_form_KKT_m"]
    rhs = np.random.default_rng(1).standard_normal(n + m)
    # polish=1: raw KKT solution with delta on the (2,2) block (qdldl_interface.c:254-265, :563-565)
    s = R.HipLDLSolver(P, A, 1e-3, None, polish=1)
    assert s.status == 0
    x = s.solve(rhs)
    K = full_kkt(P, A, 1e-3, np.full(m, 1e3))
    assert np.max(np.abs(K @ x - rhs)) < 1e-9
    s.free()
    # update_matrices / update_rho_vec through the vtable == fresh oracle factorisation
    rho = np.full(m, d["test_form_KKT_rho"])
    s = R.HipLDLSolver(P, A, d["test_form_KKT_sigma"], rho)
    assert s.update_matrices(d["test_form_KKT_Pu_new"], d["test_form_KKT_A_new"]) == 0
    rho2 = np.linspace(0.3, 3.0, m)
    assert s.update_rho_vec(rho2) == 0
    x = s.solve(rhs)
    o = ob.OracleLinsys(d["test_form_KKT_Pu_new"], d["test_form_KKT_A_new"], d["test_form_KKT_sigma"], rho2)
    assert relerr(x, o.solve(rhs)) < 1e-10
    s.free()


def test_non_cvx_init_error(R):
    """tests/non_cvx/test_non_cvx.h:31-36: fewer than n positive pivots -> OSQP_NONCVX_ERROR (5), handle NULL."""
    d = load_golden("non_cvx")
    s = R.HipLDLSolver(d["P"], d["A"], 1e-6, np.full(d["m"], 0.1))
    assert s.status == 5
    s2 = R.HipLDLSolver(d["P"], d["A"], float(d["sols"]["sigma_new"]), np.full(d["m"], 0.1))
    assert s2.status == 0
    s2.free()


# ------------------------------------------------------------------ batched factor / solve / updates ----
@pytest.mark.parametrize("shape", [(20, 35, 0.2, 5), (50, 100, 0.15, 1000), (7, 0, 0.4, 3), (1, 1, 1.0, 9)])
def test_batched_factor_and_solve_match_oracle(R, shape):
    n, m, dens, pseed = shape
    wl = R.workloads.SharedPatternQPs(n=n, m=m, density=dens, pattern_seed=pseed)
    B = 6
    Px, Ax, q, l, u = wl.values(B)
    rho = 0.05 + np.random.default_rng(0).random((B, m))
    ls = R.BatchLinsys(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), 1e-6, dev(rho))
    assert ls.status == 0
    sym = ls.export_symbolic()
    st = ls.factor_status()
    assert (st == n).all()
    rhs = np.random.default_rng(1).standard_normal((B, n + m))
    sol = ls.solve(dev(rhs)).cpu().numpy()
    for b in range(B):
        P, qq, A, ll, uu = wl.instance(b)
        o = ob.OracleLinsys(P, A, 1e-6, rho[b], perm=sym["perm"])
        e = o.export()
        for k in ("etree", "Lnz", "Lp", "Li"):
            assert (e[k] == sym[k]).all(), k                      # integers: bit-exact
        f = ls.export_factor(b)
        assert relerr(f["Lx"], e["Lx"]) < 1e-12
        assert np.max(np.abs(f["D"] - e["D"]) / np.abs(e["D"])) < 1e-12
        assert np.max(np.abs(f["Dinv"] - e["Dinv"]) / np.abs(e["Dinv"])) < 1e-12
        assert relerr(sol[b], o.solve(rhs[b])) < 1e-10
        # identity P K P' = L D L' (quantities no reference test inspects)
        N = n + m
        K = full_kkt(P, A, 1e-6, rho[b]) if m else (sparse.csc_matrix(P) + sparse.triu(P, 1).T).toarray() + 1e-6 * np.eye(n)
        Kp = K[np.ix_(sym["perm"], sym["perm"])]
        Ld = dense_from_L(sym["Lp"], sym["Li"], f["Lx"], N)
        assert np.max(np.abs(Ld @ np.diag(f["D"]) @ Ld.T - Kp)) <= 1e-12 * max(1.0, np.max(np.abs(K)))
    ls.free()


@pytest.mark.parametrize("shape", [(6, 10, 0.5, 11), (12, 20, 0.3, 12), (16, 30, 0.25, 13), (24, 40, 0.2, 14), (33, 60, 0.18, 15),
                                   (40, 90, 0.15, 16), (48, 110, 0.15, 17), (56, 120, 0.14, 18), (62, 130, 0.13, 19)])
def test_tail_sizes_across_the_kernel_instantiations(R, shape):
    """The arrowhead kernels are instantiated per tail size (16 / 32 / 48 / 56 / 64: k_arrow_factor with the inverse of the tail on the
    matrix cores in 1 - 4 blocks of 16, k_tile_solve3's tile grid): problems whose dense tail lands in every one of them, factor and
    solve against the oracle; a tail beyond 64 falls back to the generic kernels and must give the same answers."""
    n, m, dens, pseed = shape
    wl = R.workloads.SharedPatternQPs(n=n, m=m, density=dens, pattern_seed=pseed)
    B = 5
    Px, Ax, q, l, u = wl.values(B)
    rho = 0.05 + np.random.default_rng(2).random((B, m))
    ls = R.BatchLinsys(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), 1e-6, dev(rho))
    assert ls.status == 0 and (ls.factor_status() == n).all()
    sym = ls.export_symbolic()
    rhs = np.random.default_rng(3).standard_normal((B, n + m))
    sol = ls.solve(dev(rhs)).cpu().numpy()
    again = ls.solve(dev(rhs)).cpu().numpy()
    assert np.array_equal(sol, again)
    for b in range(B):
        P, qq, A, ll, uu = wl.instance(b)
        o = ob.OracleLinsys(P, A, 1e-6, rho[b], perm=sym["perm"])
        assert relerr(ls.export_factor(b)["Lx"], o.export()["Lx"]) < 1e-12
        assert relerr(sol[b], o.solve(rhs[b])) < 1e-10
    # new values through update_matrices (the refactorisation path of the benchmark): same check
    Px2, Ax2 = wl.values(B, seed0=77)[:2]
    assert ls.update_matrices(dev(Px2), dev(Ax2)) == 0
    sol2 = ls.solve(dev(rhs)).cpu().numpy()
    for b in range(B):
        P2, _, A2, _, _ = wl.instance(b, seed0=77)
        o = ob.OracleLinsys(P2, A2, 1e-6, rho[b], perm=sym["perm"])
        assert relerr(sol2[b], o.solve(rhs[b])) < 1e-10
    ls.free()


def test_batched_update_rho_vec_and_matrices_match_fresh_factorisation(R):
    wl = R.workloads.SharedPatternQPs(n=20, m=35, density=0.2, pattern_seed=11)
    B = 5
    Px, Ax, q, l, u = wl.values(B)
    Px2, Ax2, _, _, _ = wl.values(B, seed0=100)
    rho = np.full((B, wl.m), 0.1)
    rho2 = 0.05 + np.random.default_rng(3).random((B, wl.m))
    ls = R.BatchLinsys(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), 1e-6, dev(rho))
    sym = ls.export_symbolic()
    assert ls.update_rho_vec(dev(rho2)) == 0
    fresh = R.BatchLinsys(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), 1e-6, dev(rho2))
    for b in range(B):
        assert np.array_equal(ls.export_factor(b)["Lx"], fresh.export_factor(b)["Lx"])   # same kernel, same inputs
    # masked update: only instances 1 and 3 move
    mask = torch.tensor([0, 1, 0, 1, 0], dtype=torch.int32, device="cuda:0")
    before = [ls.export_factor(b)["Lx"].copy() for b in range(B)]
    assert ls.update_rho_vec(dev(rho), mask) == 0
    for b in range(B):
        now = ls.export_factor(b)["Lx"]
        if b in (1, 3):
            P, qq, A, ll, uu = wl.instance(b)
            o = ob.OracleLinsys(P, A, 1e-6, rho[b], perm=sym["perm"])
            assert relerr(now, o.export()["Lx"]) < 1e-12
        else:
            assert np.array_equal(now, before[b])
    # update_matrices (qdldl_interface.c:590-602): new P and A values, same pattern
    assert ls.update_rho_vec(dev(rho)) == 0
    assert ls.update_matrices(dev(Px2), dev(Ax2)) == 0
    rhs = np.random.default_rng(5).standard_normal((B, wl.n + wl.m))
    sol = ls.solve(dev(rhs)).cpu().numpy()
    for b in range(B):
        P, qq, A, ll, uu = wl.instance(b, seed0=100)
        o = ob.OracleLinsys(P, A, 1e-6, rho[b], perm=sym["perm"])
        assert relerr(ls.export_factor(b)["Lx"], o.export()["Lx"]) < 1e-12
        assert relerr(sol[b], o.solve(rhs[b])) < 1e-10
    ls.free(); fresh.free()


def test_batch_with_one_non_convex_instance_reports_it(R):
    wl = R.workloads.SharedPatternQPs(n=10, m=12, density=0.3, pattern_seed=2)
    Px, Ax, q, l, u = wl.values(4)
    Px[2] = -Px[2]                       # indefinite P for instance 2
    h = R.BatchLinsys(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), 1e-6, dev(np.full((4, wl.m), 0.1)))
    assert h.status == 5                 # RLDL_NONCVX_ERROR, handle not created (qdldl_interface.c:298-303)


def test_user_permutation_and_large_problem_global_memory_path(R):
    """nnzL + N above the LDS budget exercises the global-memory variants of factor / solve."""
    wl = R.workloads.MPCStageQPs(N=20)
    B = 3
    Px, Ax, q, l, u = wl.values(B)
    rho = np.full((B, wl.m), 0.1)
    perm = R.workloads.stage_permutation(*wl.dims)
    ls = R.BatchLinsys(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), 1e-6, dev(rho), perm=perm)
    assert ls.status == 0
    d = ls.dims()
    assert 8 * (d["nnzL"] + d["n"] + d["m"]) > 64 * 1024
    sym = ls.export_symbolic()
    assert (sym["perm"] == perm).all()
    rhs = np.random.default_rng(2).standard_normal((B, wl.n + wl.m))
    sol = ls.solve(dev(rhs)).cpu().numpy()
    for b in range(B):
        P, qq, A, ll, uu = wl.instance(b)
        o = ob.OracleLinsys(P, A, 1e-6, rho[b], perm=perm)
        assert relerr(ls.export_factor(b)["Lx"], o.export()["Lx"]) < 1e-11
        assert relerr(sol[b], o.solve(rhs[b])) < 1e-9
    ls.free()


# ------------------------------------------------------------------ stage-recursive strategy ----
def test_recursive_init_uses_closed_form_permutation_and_restart_equals_refactor(R):
    wl = R.workloads.MPCStageQPs(N=6)
    B = 4
    Px, Ax, q, l, u = wl.values(B)
    rho = np.full((B, wl.m), 0.1)
    ls = R.BatchLinsys.recursive(wl.dims, wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), 1e-6, dev(rho))
    assert ls.status == 0
    sym = ls.export_symbolic()
    assert (sym["perm"] == R.workloads.stage_permutation(*wl.dims)).all()      # bit-exact integers
    for b in range(B):
        P, qq, A, ll, uu = wl.instance(b)
        K = full_kkt(P, A, 1e-6, rho[b])
        Kp = K[np.ix_(sym["perm"], sym["perm"])]
        f = ls.export_factor(b)
        Ld = dense_from_L(sym["Lp"], sym["Li"], f["Lx"], wl.n + wl.m)
        assert np.max(np.abs(Ld @ np.diag(f["D"]) @ Ld.T - Kp)) <= 1e-11 * np.max(np.abs(K))
    # config 5: perturb P, A of stages >= k only; restart-from-stage must equal a full refactor
    k = 3
    rg = np.random.default_rng(7)
    Px2 = Px * np.where(wl.P_stage >= k, 1 + 0.05 * rg.standard_normal(Px.shape), 1.0)
    Ax2 = Ax * np.where((wl.A_stage >= k) & (Ax != -1.0), 1 + 0.05 * rg.standard_normal(Ax.shape), 1.0)
    assert ls.update_from_stage(k, dev(Px2), dev(Ax2), None) == 0
    full = R.BatchLinsys.recursive(wl.dims, wl.P_pattern, wl.A_pattern, dev(Px2), dev(Ax2), 1e-6, dev(rho))
    rhs = np.random.default_rng(3).standard_normal((B, wl.n + wl.m))
    s1 = ls.solve(dev(rhs)).cpu().numpy(); s2 = full.solve(dev(rhs)).cpu().numpy()
    for b in range(B):
        assert relerr(ls.export_factor(b)["Lx"], full.export_factor(b)["Lx"]) < 1e-12
        assert relerr(ls.export_factor(b)["D"], full.export_factor(b)["D"]) < 1e-12
    assert relerr(s1, s2) < 1e-10
    # a rho change touches every constraint pivot: restart from stage 0 == update_rho_vec
    rho2 = np.full((B, wl.m), 0.4)
    assert ls.update_from_stage(0, None, None, dev(rho2)) == 0
    assert full.update_rho_vec(dev(rho2)) == 0
    for b in range(B):
        assert relerr(ls.export_factor(b)["Lx"], full.export_factor(b)["Lx"]) < 1e-12
    ls.free(); full.free()


# ------------------------------------------------------------------ ADMM driver ----
FIXED = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=200, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)


def test_admm_200_fixed_iterations_match_oracle(R):
    """Config 2 at oracle-sized batch: factor + 200 ADMM iterations, identical rho schedule."""
    wl = R.workloads.SharedPatternQPs()           # n=50, m=100, density 0.15
    B = 6
    Px, Ax, q, l, u = wl.values(B)
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), **FIXED)
    assert w.status == 0
    r = w.solve()
    perm = w.linsys().export_symbolic()["perm"]
    for b in range(B):
        P, qq, A, ll, uu = wl.instance(b)
        ro = ob.OracleOSQP(P, qq, A, ll, uu, perm=perm, **FIXED).solve()
        assert relerr(r["x"][b].cpu().numpy(), ro["x_iter"]) < 1e-8
        assert relerr(r["y"][b].cpu().numpy(), ro["y_iter"]) < 1e-8
        assert relerr(r["z"][b].cpu().numpy(), ro["z_iter"]) < 1e-8
        assert int(r["status"][b]) == ro["status"] and int(r["iter"][b]) == ro["iter"] == 200
        assert abs(float(r["pri_res"][b]) - ro["pri_res"]) < 1e-8 * max(1, ro["pri_res"])
        assert abs(float(r["dua_res"][b]) - ro["dua_res"]) < 1e-8 * max(1, ro["dua_res"])
        assert abs(float(r["obj"][b]) - ro["obj"]) < 1e-8 * max(1, abs(ro["obj"]))
    w.cleanup()


def test_admm_termination_and_adaptive_rho_match_oracle(R):
    wl = R.workloads.SharedPatternQPs(n=20, m=30, density=0.25, pattern_seed=21)
    B = 8
    Px, Ax, q, l, u = wl.values(B)
    l[:, :4] = u[:, :4] = 0.5 * (l[:, :4] + u[:, :4])          # some equality rows -> rho_vec is not uniform
    u[:, 4] = 1e30; l[:, 5] = -1e30; l[:, 6] = -1e30; u[:, 6] = 1e30   # one-sided and free rows
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=4000, check_termination=25, adaptive_rho=1,
              adaptive_rho_interval=50, eps_abs=1e-5, eps_rel=1e-5, warm_start=0, scaling=0)
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), **kw)
    r = w.solve()
    perm = w.linsys().export_symbolic()["perm"]
    for b in range(B):
        P, qq, A, ll, uu = wl.instance(b)
        ll = l[b]; uu = u[b]
        ro = ob.OracleOSQP(P, qq, A, ll, uu, perm=perm, **kw).solve()
        assert int(r["status"][b]) == ro["status"] == 1
        assert int(r["iter"][b]) == ro["iter"], (b, int(r["iter"][b]), ro["iter"])
        assert relerr(r["x"][b].cpu().numpy(), ro["x"]) < 1e-7
        assert relerr(r["y"][b].cpu().numpy(), ro["y"]) < 1e-7
    w.cleanup()


def _solve_golden(R, P, q, A, l, u, reps=3, **kw):
    """Run a reference fixture as a batch of `reps` identical instances."""
    Pc, Ac = R.CscPattern(P, upper=True), R.CscPattern(A)
    base = dict(eps_abs=1e-7, eps_rel=1e-7, max_iter=20000, check_termination=1, scaling=0, adaptive_rho=1,
                adaptive_rho_interval=25)
    base.update(kw)
    tile = lambda v: dev(np.tile(np.asarray(v, float), (reps, 1)))
    w = R.OSQPBatch(Pc, Ac, tile(Pc.x), tile(Ac.x), tile(q), tile(osqp_inf(l)), tile(osqp_inf(u)), **base)
    return w


def test_basic_qp_golden_and_updates(R):
    d = load_golden("basic_qp"); s = d["sols"]
    w = _solve_golden(R, d["P"], d["q"], d["A"], d["l"], d["u"])
    assert w.status == 0
    r = w.solve()
    for b in range(3):
        assert int(r["status"][b]) == 1
        assert np.max(np.abs(r["x"][b].cpu().numpy() - s["x_test"])) < TESTS_TOL
        assert np.max(np.abs(r["y"][b].cpu().numpy() - s["y_test"])) < TESTS_TOL
        assert abs(float(r["obj"][b]) - s["obj_value_test"]) < TESTS_TOL
    # osqp_update_lin_cost + osqp_update_bounds (test_basic_qp.h:88-240) vs a fresh oracle solve
    tile = lambda v: dev(np.tile(np.asarray(v, float), (3, 1)))
    assert w.update_lin_cost(tile(s["q_new"])) == 0
    assert w.update_bounds(tile(osqp_inf(s["l_new"])), tile(osqp_inf(s["u_new"]))) == 0
    r = w.solve()
    ro = ob.OracleOSQP(d["P"], s["q_new"], d["A"], osqp_inf(s["l_new"]), osqp_inf(s["u_new"]), eps_abs=1e-7, eps_rel=1e-7,
                       max_iter=20000, check_termination=1, scaling=0, adaptive_rho=1, adaptive_rho_interval=25).solve()
    assert int(r["status"][0]) == ro["status"] == 1
    assert np.max(np.abs(r["x"][0].cpu().numpy() - ro["x"])) < TESTS_TOL
    assert np.max(np.abs(r["y"][0].cpu().numpy() - ro["y"])) < TESTS_TOL
    w.cleanup()


def test_basic_qp_update_rho_same_iteration_count(R):
    """tests/basic_qp/test_basic_qp.h:651-779."""
    d = load_golden("basic_qp")
    kw = dict(adaptive_rho=0, eps_abs=5e-5, eps_rel=5e-5, check_termination=1, max_iter=4000)
    a = _solve_golden(R, d["P"], d["q"], d["A"], d["l"], d["u"], rho=0.7, **kw)
    ra = a.solve()
    b = _solve_golden(R, d["P"], d["q"], d["A"], d["l"], d["u"], rho=0.1, warm_start=0, **kw)
    b.solve()
    assert b.update_rho(0.7) == 0
    rb = b.solve()
    assert int(ra["iter"][0]) == int(rb["iter"][0]) > 0
    assert np.max(np.abs(ra["x"][0].cpu().numpy() - rb["x"][0].cpu().numpy())) < 1e-9
    a.cleanup(); b.cleanup()


def test_update_matrices_golden(R):
    """tests/update_matrices/test_update_matrices.h:74-313 (osqp_update_P / _A / _P_A known answers)."""
    d = load_golden("update_matrices")["data"]
    w = _solve_golden(R, d["test_solve_Pu"], d["test_solve_q"], d["test_solve_A"], d["test_solve_l"], d["test_solve_u"], reps=2)
    r = w.solve()
    assert int(r["status"][0]) == 1
    assert np.max(np.abs(r["x"][0].cpu().numpy() - d["test_solve_x"])) < TESTS_TOL
    assert abs(float(r["obj"][0]) - d["test_solve_obj_value"]) < TESTS_TOL
    Pn = sparse.csc_matrix(d["test_solve_Pu_new"]); Pn.sort_indices()
    An = sparse.csc_matrix(d["test_solve_A_new"]); An.sort_indices()
    tile = lambda v: dev(np.tile(np.asarray(v, float), (2, 1)))
    assert w.update_P_A(Px=tile(Pn.data)) == 0
    r = w.solve()
    assert np.max(np.abs(r["x"][1].cpu().numpy() - d["test_solve_P_new_x"])) < TESTS_TOL
    assert abs(float(r["obj"][1]) - d["test_solve_P_new_obj_value"]) < TESTS_TOL
    assert w.update_P_A(Px=tile(Pn.data), Ax=tile(An.data)) == 0
    r = w.solve()
    assert np.max(np.abs(r["x"][0].cpu().numpy() - d["test_solve_P_A_new_x"])) < TESTS_TOL
    assert abs(float(r["obj"][0]) - d["test_solve_P_A_new_obj_value"]) < TESTS_TOL
    w.cleanup()


def test_unconstrained_and_infeasibility_golden(R):
    d = load_golden("unconstrained"); s = d["sols"]
    w = _solve_golden(R, d["P"], d["q"], d["A"], d["l"], d["u"], reps=2)
    r = w.solve()
    assert int(r["status"][0]) == 1
    assert np.max(np.abs(r["x"][0].cpu().numpy() - s["x_test"])) < TESTS_TOL
    assert abs(float(r["obj"][0]) - s["obj_value_test"]) < TESTS_TOL
    w.cleanup()
    g = load_golden("primal_dual_infeasibility")["data"]
    cases = [("A12", "u1", (1,)), ("A12", "u2", (-3,)), ("A34", "u3", (-4,)), ("A34", "u4", (-3, -4))]
    for Ak, uk, ok in cases:
        w = _solve_golden(R, g["P"], g["q"], g[Ak], g["l"], g[uk], reps=2, eps_abs=1e-6, eps_rel=1e-6, max_iter=2000,
                          check_termination=25)
        r = w.solve()
        assert int(r["status"][0]) in ok, (Ak, uk, int(r["status"][0]))
        if ok == (1,):
            assert np.max(np.abs(r["x"][0].cpu().numpy() - g["x1"])) < TESTS_TOL
            assert np.max(np.abs(r["y"][0].cpu().numpy() - g["y1"])) < TESTS_TOL
            assert abs(float(r["obj"][0]) - g["obj_value1"]) < TESTS_TOL
        w.cleanup()
    # non-convex divergence -> OSQP_NON_CVX with obj == OSQP_NAN (test_non_cvx.h:53-58)
    d = load_golden("non_cvx")
    w = _solve_golden(R, d["P"], d["q"], d["A"], d["l"], d["u"], reps=2, sigma=float(d["sols"]["sigma_new"]),
                      eps_abs=1e-3, eps_rel=1e-3, max_iter=4000, check_termination=25, adaptive_rho_interval=100)
    r = w.solve()
    assert int(r["status"][0]) == -7 and float(r["obj"][0]) == float(0x7fc00000)
    w.cleanup()


def test_warm_start_converges_immediately(R):
    wl = R.workloads.SharedPatternQPs(n=15, m=20, density=0.3, pattern_seed=4)
    B = 4
    Px, Ax, q, l, u = wl.values(B)
    kw = dict(eps_abs=1e-6, eps_rel=1e-6, check_termination=1, adaptive_rho=0, scaling=0, max_iter=20000)
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), **kw)
    r = w.solve()
    assert (r["status"] == 1).all()
    it0 = r["iter"].clone()
    w2 = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), **kw)
    assert w2.warm_start(r["x"], r["y"]) == 0
    r2 = w2.solve()
    assert (r2["status"] == 1).all() and (r2["iter"] <= torch.clamp(it0 // 4, min=2)).all()
    w.cleanup(); w2.cleanup()


# ------------------------------------------------------------------ full-size properties (BASELINE sizes) ----
def test_full_size_batch_4096_properties(R):
    """Metric shape n=50, m=100, batch=4096: size-independent properties instead of an oracle sweep."""
    wl = R.workloads.SharedPatternQPs()
    B = 4096
    Px, Ax, q, l, u = wl.values(B)
    rho = np.full((B, wl.m), 0.1)
    dPx, dAx = dev(Px), dev(Ax)
    ls = R.BatchLinsys(wl.P_pattern, wl.A_pattern, dPx, dAx, 1e-6, dev(rho))
    assert ls.status == 0 and (ls.factor_status() == wl.n).all()
    N = wl.n + wl.m
    g = torch.Generator(device="cuda:0"); g.manual_seed(0)
    rhs = torch.randn((B, N), dtype=torch.float64, device="cuda:0", generator=g)
    sol = ls.solve(rhs.clone())
    # (1) linearity: solve(a*r1 + r2) == a*solve(r1) + solve(r2) on the x-part
    r2 = torch.randn((B, N), dtype=torch.float64, device="cuda:0", generator=g)
    s2 = ls.solve(r2.clone())
    s3 = ls.solve((2.5 * rhs + r2).clone())
    assert float((s3[:, :wl.n] - (2.5 * sol[:, :wl.n] + s2[:, :wl.n])).abs().max()) < 1e-7
    # (2) residual K * raw = rhs on a spread sample of instances (raw nu recovered from the epilogue)
    for b in range(0, B, 257):
        P, qq, A, ll, uu = wl.instance(b)
        K = full_kkt(P, A, 1e-6, rho[b])
        out = sol[b].cpu().numpy(); r = rhs[b].cpu().numpy()
        raw = np.concatenate([out[:wl.n], (out[wl.n:] - r[wl.n:]) * rho[b]])
        assert np.max(np.abs(K @ raw - r)) < 1e-8 * max(1.0, np.max(np.abs(raw)))
    # (3) refactor idempotence: update_matrices with the same values reproduces L bit for bit
    f0 = ls.export_factor(1234)["Lx"].copy()
    assert ls.update_matrices(dPx, dAx) == 0
    assert np.array_equal(ls.export_factor(1234)["Lx"], f0)
    # (4) every instance of the batch equals the same instance solved in a batch of its own
    one = R.BatchLinsys(wl.P_pattern, wl.A_pattern, dPx[4095:4096].contiguous(), dAx[4095:4096].contiguous(), 1e-6,
                        dev(rho[4095:4096]))
    assert np.array_equal(one.export_factor(0)["Lx"], ls.export_factor(4095)["Lx"])
    ls.free(); one.free()
    # (5) fused ADMM at full batch: all instances reach the fixed iteration count and identical
    #     instances give identical results (batch position must not matter)
    qd, ld, ud = dev(q), dev(l), dev(u)
    dPx[1] = dPx[0]; dAx[1] = dAx[0]; qd[1] = qd[0]; ld[1] = ld[0]; ud[1] = ud[0]
    dPx[4095] = dPx[0]; dAx[4095] = dAx[0]; qd[4095] = qd[0]; ld[4095] = ld[0]; ud[4095] = ud[0]
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dPx, dAx, qd, ld, ud, **FIXED)
    r = w.solve()
    assert (r["iter"] == 200).all()
    assert torch.equal(r["x"][0], r["x"][1]) and torch.equal(r["x"][0], r["x"][4095])
    assert bool(torch.isfinite(r["x"]).all())
    w.cleanup()


def test_config4_shard_8192_position_independence_across_rounds(R):
    """BASELINE config 4's per-GPU shard (65536 / 8 = 8192 instances of the metric shape): the tile kernels hold 2048 waves at a
    time, so 8192 instances are four resident rounds.  Copies of one instance placed in every round (and at both ends of a
    round) must give bit-identical factors, solves and iterates; a spread sample agrees with the oracle."""
    wl = R.workloads.SharedPatternQPs()
    B = 8192
    Px, Ax, q, l, u = wl.values(B)
    spots = [1, 2047, 2048, 4095, 4096, 6143, 6144, 8191]          # first / last wave of every round
    for arr in (Px, Ax, q, l, u):
        arr[spots] = arr[0]
    dPx, dAx = dev(Px), dev(Ax)
    rho = np.full((B, wl.m), 0.1)
    ls = R.BatchLinsys(wl.P_pattern, wl.A_pattern, dPx, dAx, 1e-6, dev(rho))
    assert ls.status == 0 and (ls.factor_status() == wl.n).all()
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    rhs = torch.randn((B, wl.n + wl.m), dtype=torch.float64, device="cuda:0", generator=g)
    rhs[spots] = rhs[0].clone()
    sol = ls.solve(rhs.clone())
    f0 = ls.export_factor(0)["Lx"]
    for k in spots:
        assert np.array_equal(ls.export_factor(k)["Lx"], f0) and torch.equal(sol[k], sol[0]), k
    sym = ls.export_symbolic()
    for b in (0, 3000, 5555, 8190):
        P, qq, A, ll, uu = wl.instance(b) if b not in spots else wl.instance(0)
        ref = ob.OracleLinsys(P, A, 1e-6, rho[b], perm=sym["perm"]).solve(rhs[b].cpu().numpy())
        assert relerr(sol[b].cpu().numpy(), ref) < 1e-10
    ls.free()
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dPx, dAx, dev(q), dev(l), dev(u), **FIXED)
    r = w.solve()
    assert (r["iter"] == 200).all() and bool(torch.isfinite(r["x"]).all())
    for k in spots:
        assert torch.equal(r["x"][k], r["x"][0]) and torch.equal(r["y"][k], r["y"][0]), k
    for b in (4097, 8190):
        P, qq, A, ll, uu = wl.instance(b)
        ro = ob.OracleOSQP(P, qq, A, ll, uu, perm=sym["perm"], **FIXED).solve()
        assert relerr(r["x_iter"][b].cpu().numpy(), ro["x_iter"]) < 1e-8
    w.cleanup()


def test_solve_wave_timeline_and_rotating_timing(R):
    """Tracing aids of the plugin solve (rldl_batch_trace_solve, rldl_batch_time_solve_rotating): the timeline of a launch is ordered
    per wave, and the traced / rotated launches are ordinary solves (same result as rldl_batch_solve)."""
    wl = R.workloads.SharedPatternQPs()
    B = 64
    Px, Ax, q, l, u = wl.values(B)
    rho = np.full((B, wl.m), 0.1)
    hs = [R.BatchLinsys(wl.P_pattern, wl.A_pattern, dev(Px * (1.0 + 0.01 * k)), dev(Ax), 1e-6, dev(rho)) for k in range(3)]
    g = torch.Generator(device="cuda:0"); g.manual_seed(3)
    rhs = torch.randn((B, wl.n + wl.m), dtype=torch.float64, device="cuda:0", generator=g)
    ref = hs[0].solve(rhs.clone())
    tr = hs[0].trace_solve(rhs.clone())
    if tr is not None:                                              # (None: a kernel-selection switch took the handle off the tile kernel)
        assert tr.shape == (B, 8) and (tr[:, :7] > 0).all()
        assert (np.diff(tr[:, [0, 7, 1, 2, 3, 4, 5, 6]], axis=1) >= 0).all()      # start, loads issued, landed, gather, forward, backward, scatter, stores
    hs[1].set_cache_policy("stream")                               # non-temporal row loads: same arithmetic, same result
    nt = hs[1].solve(rhs.clone())
    hs[1].set_cache_policy("resident")
    assert torch.equal(nt, hs[1].solve(rhs.clone()))
    hs[1].set_cache_policy("auto")
    bs = [rhs.clone() for _ in hs]
    assert R.BatchLinsys.time_solve_rotating(hs, bs, reps=3) > 0.0                  # one launch per handle: bs[k] = solve of handle k
    assert torch.equal(bs[0], ref)
    for k in (1, 2):
        assert torch.equal(bs[k], hs[k].solve(rhs.clone()))
    # the factor kernel's timeline: an ordinary refactorisation (factor and solve unchanged), stamps ordered per wave
    tf = hs[0].trace_factor()
    if tf is not None:
        fused = bool((tf[:, 6:] > 0).all())                         # (RLDL_SPLIT_INVERT: the tail inverse is its own launch, stamps 6 and 7 stay 0)
        tf = tf if fused else tf[:, :6]
        assert tf.shape[0] == B and (tf > 0).all() and (np.diff(tf, axis=1) >= 0).all()
        assert torch.equal(hs[0].solve(rhs.clone()), ref)
    for h in hs:
        h.free()


def test_warm_started_fixed_iteration_solves_continue_where_the_last_one_stopped(R):
    """Without termination checks the first launch of the tile kernel starts the solve itself (status, cold start: no k_solve_begin
    launch).  Cold: 50 iterations.  Warm (osqp_solve with warm_start = 1, osqp.c:380-385 keeps x, z, y): 30 iterations, then 20 more
    from where they stopped -- the same 50 iterations, bit for bit; and a cold solve after them starts from zero again."""
    wl = R.workloads.SharedPatternQPs()
    B = 7
    Px, Ax, q, l, u = wl.values(B)
    base = dict(rho=0.1, sigma=1e-6, alpha=1.6, check_termination=0, adaptive_rho=0, scaling=0)
    wc = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), max_iter=50, warm_start=0, **base)
    ww = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), max_iter=30, warm_start=1, **base)
    r50 = wc.solve()
    r30 = ww.solve()
    assert not torch.equal(r30["x"], r50["x"])
    assert ww.update_settings(max_iter=20) == 0
    r30_20 = ww.solve()
    assert torch.equal(r30_20["x"], r50["x"]) and torch.equal(r30_20["y"], r50["y"]) and torch.equal(r30_20["z"], r50["z"])
    assert (r30_20["status"] == r50["status"]).all()
    again = wc.solve()                                            # cold start: the iterates of the last solve do not leak in
    assert torch.equal(again["x"], r50["x"]) and torch.equal(again["y"], r50["y"])
    wc.cleanup(); ww.cleanup()


def test_resident_iterations_equal_single_iteration_launches(R):
    """The fused kernel keeps an instance's factor and iterates on chip for a whole group of iterations; running the
    same number of iterations as separate one-iteration launches (state through HBM every time) must give the same
    iterates bit for bit, and a solve split by termination checks must match the oracle's iteration count."""
    wl = R.workloads.SharedPatternQPs()
    B = 8
    Px, Ax, q, l, u = wl.values(B)
    K = 37
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=K, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
    a = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), **kw)
    ra = a.solve()
    assert a.last_loop()[1] == K and a.last_loop()[2] == 1          # one launch for the whole group
    b = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), **kw)
    b.time_iteration(K)                                               # K launches of one iteration each, from the cold start
    rb = b.results()
    for key in ("x_iter", "y_iter", "z"):
        assert torch.equal(ra[key], rb[key]), key
    a.cleanup(); b.cleanup()


def test_arrowhead_factor_inertia_and_odd_batch(R):
    """k_arrow_factor on the metric shape: status = number of positive pivots (qdldl_interface.c:80-92) per instance,
    a non-convex instance is reported, and a batch that is neither a multiple of the workgroup size nor within one
    resident round (4099 > 16 waves x 256 CUs) gives identical answers for identical instances."""
    wl = R.workloads.SharedPatternQPs()
    B = 6
    Px, Ax, q, l, u = wl.values(B)
    rho = np.full((B, wl.m), 0.1)
    ls = R.BatchLinsys(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), 1e-6, dev(rho))
    assert ls.status == 0 and (ls.factor_status() == wl.n).all()
    Px2 = Px.copy(); Px2[4] = -Px2[4]                               # indefinite P for instance 4 only
    assert ls.update_matrices(dev(Px2), dev(Ax)) == 1                # reference: update returns non-zero (qdldl_interface.c:598-600)
    st = ls.factor_status()
    assert st[4] < wl.n and (np.delete(st, 4) == wl.n).all()
    ls.free()
    # 4099 copies of two instances, interleaved
    Bb = 4099
    sel = np.arange(Bb) % 2
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=30, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px[sel]), dev(Ax[sel]), dev(q[sel]), dev(l[sel]), dev(u[sel]), **kw)
    r = w.solve()
    for k in (0, 1):
        idx = torch.from_numpy(np.nonzero(sel == k)[0]).cuda()
        xs = r["x"][idx]
        assert torch.equal(xs, xs[:1].expand_as(xs))
    ro = ob.OracleOSQP(*[wl.instance(1)[i] for i in (0, 1, 2, 3, 4)], perm=w.linsys().export_symbolic()["perm"], **kw).solve()
    assert relerr(r["x"][1].cpu().numpy(), ro["x"]) < 1e-8
    w.cleanup()


def test_pattern_groups_solve_mixed_sparsity_batches(R):
    """Instances with different sparsity patterns: bucketed by pattern, one workspace and stream per bucket, enqueued
    together (osqp_batch_solve_async), results back in the caller's order and equal to the oracle per instance."""
    wls = [R.workloads.SharedPatternQPs(n=20, m=30, density=0.25, pattern_seed=s) for s in (3, 4, 5)]
    problems, src = [], []
    for k in range(14):                                            # interleave the three patterns
        wl = wls[k % 3]
        problems.append(wl.instance(k // 3))
        src.append((k % 3, k // 3))
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=60, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
    g = R.OSQPBatchGroups(problems, **kw)
    assert g.n_patterns == 3
    r = g.solve()
    for k, (P, q, A, l, u) in enumerate(problems):
        w = [w for idx, w in g.groups if int(k) in idx.tolist()][0]
        ro = ob.OracleOSQP(P, q, A, l, u, perm=w.linsys().export_symbolic()["perm"], **kw).solve()
        assert relerr(r["x"][k].cpu().numpy(), ro["x"]) < 1e-8 and relerr(r["y"][k].cpu().numpy(), ro["y"]) < 1e-8
        assert int(r["iter"][k]) == 60
    # with termination checks the groups fall back to the blocking solve and still agree with the oracle
    kw2 = dict(kw, max_iter=2000, check_termination=10, eps_abs=1e-5, eps_rel=1e-5)
    g2 = R.OSQPBatchGroups(problems, **kw2)
    r2 = g2.solve()
    P, q, A, l, u = problems[4]
    w = [w for idx, w in g2.groups if 4 in idx.tolist()][0]
    ro = ob.OracleOSQP(P, q, A, l, u, perm=w.linsys().export_symbolic()["perm"], **kw2).solve()
    assert int(r2["iter"][4]) == ro["iter"] and int(r2["status"][4]) == ro["status"]
    g.cleanup(); g2.cleanup()


def test_pattern_groups_in_one_launch_chain(R):
    """Metric-shape instances of several sparsity patterns: when the patterns select the same kernel instantiation the solve of all
    groups is one launch chain over the stacked instances (osqp_multi_*); results equal the per-stream path bit for bit (the same
    kernels run on the same data) and the oracle per instance."""
    seeds = (2000, 2001, 2002, 2003)
    wls = [R.workloads.SharedPatternQPs(pattern_seed=s) for s in seeds]
    problems = [wls[k % 4].instance(k // 4) for k in range(4 * 9 + 2)]            # interleaved, ragged group sizes (10, 10, 9, 9)
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=40, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
    g1 = R.OSQPBatchGroups(problems, **kw)
    g0 = R.OSQPBatchGroups(problems, one_launch=False, **kw)
    assert g1.n_patterns == 4 and not g0.one_launch
    r1 = {k: v.clone() for k, v in g1.solve().items()}
    r1b = g1.solve()                                                                # a second solve of the same set: cold start again
    r0 = g0.solve()
    for key in ("x", "y", "z", "obj", "pri_res", "dua_res", "iter", "status"):
        assert torch.equal(r1[key], r0[key]) and torch.equal(r1b[key], r0[key]), key
    for k in (0, 5, 37):
        P, q, A, l, u = problems[k]
        w = [w for idx, w in g1.groups if int(k) in idx.tolist()][0]
        ro = ob.OracleOSQP(P, q, A, l, u, perm=w.linsys().export_symbolic()["perm"], **kw).solve()
        assert relerr(r1["x"][k].cpu().numpy(), ro["x"]) < 1e-8 and relerr(r1["y"][k].cpu().numpy(), ro["y"]) < 1e-8
    if not any(os.environ.get(k) for k in ("RLDL_NO_TILE", "RLDL_NO_ARROW", "RLDL_CHECK_STAGED")):   # (kernel-selection switches take the set off the single chain)
        assert g1.one_launch                                                       # arrowhead patterns on the tile kernels (two instantiations here)
    # new P / A values for every group: one update chain over the stacked instances (osqp_multi_update_P_A) against one asynchronous
    # update per workspace -- same scatter, factor and tail-inverse kernels on the same data, so the next solve is bit-identical too
    def values(g, fp, fa):
        out = []
        for idx, w in g.groups:
            Pu = [sparse.triu(sparse.csc_matrix(problems[i][0]), format="csc") for i in idx.tolist()]
            Ac = [sparse.csc_matrix(problems[i][2]) for i in idx.tolist()]
            for M in Pu + Ac:
                M.sort_indices()
            out.append((dev(np.stack([M.data for M in Pu]) * fp), dev(np.stack([M.data for M in Ac]) * fa)))
        return out
    g1.update_P_A(values(g1, 1.05, 0.97)); g0.update_P_A(values(g0, 1.05, 0.97))
    u1, u0 = g1.solve(), g0.solve()
    for key in ("x", "y", "z", "obj", "pri_res", "dua_res", "iter", "status"):
        assert torch.equal(u1[key], u0[key]), key
    assert not torch.equal(u1["x"], r0["x"])                                        # (the update took effect)
    P, q, A, l, u = problems[5]
    w = [w for idx, w in g1.groups if 5 in idx.tolist()][0]
    ro = ob.OracleOSQP(P * 1.05, q, A * 0.97, l, u, perm=w.linsys().export_symbolic()["perm"], **kw).solve()
    assert relerr(u1["x"][5].cpu().numpy(), ro["x"]) < 1e-8 and relerr(u1["y"][5].cpu().numpy(), ro["y"]) < 1e-8
    # a refactorisation that fails (P -> -P: wrong inertia, osqp.c:1246-1262) is reported by the next solve of the set, once
    g1.update_P_A(values(g1, -1.0, 1.0))
    with pytest.raises(RuntimeError):
        g1.solve()
    g1.update_P_A(values(g1, 1.05, 0.97))
    u2 = g1.solve()
    assert torch.equal(u2["x"], u1["x"])
    g1.cleanup(); g0.cleanup()


def test_one_sparsity_pattern_per_instance_in_one_launch_chain(R):
    """BASELINE config 2, literal reading: every instance with its own sparsity pattern (the reference: each osqp_setup analyses its own
    pattern, qdldl_interface.c:99-166).  One workspace per pattern, all of them in ONE set (more than the 64 groups one descriptor used to
    hold); patterns whose owner-gather steps fit no compiled split run the scatter variant of the fused kernel (k_tile_admm<.., TK = 0>)
    inside the same chain.  Every instance equals the oracle set up on its own pattern."""
    from osqp_recursive_ldl_amd import _lib
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=60, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
    problems, scatter = [], 0
    for s in range(400):                                            # ~1.5 % of the random patterns need the scatter variant: make sure some are in
        wl = R.workloads.SharedPatternQPs(pattern_seed=5000 + s)
        if len(problems) >= 70 and scatter >= 2:
            break
        if len(problems) >= 70:                                     # (only looking for scatter patterns from here on)
            Px, Ax, q, l, u = wl.values(1)
            w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), **kw)
            key = int(_lib.lib().osqp_batch_multi_key(w.h))
            w.cleanup()
            if key < 0 or key % 4096 != 0:
                continue
        problems.append(wl.instance(0))
    g = R.OSQPBatchGroups(problems, **kw)
    assert g.n_patterns == len(problems) > 64
    keys = [int(_lib.lib().osqp_batch_multi_key(w.h)) for _, w in g.groups]
    if not any(os.environ.get(k) for k in ("RLDL_NO_TILE", "RLDL_NO_ARROW", "RLDL_CHECK_STAGED")):
        assert g.one_launch and g.groups_outside_the_chain == 0 and min(keys) >= 0
        assert sum(1 for k in keys if k % 4096 == 0) >= 2           # scatter-variant patterns are in the chain
    r = {k: v.clone() for k, v in g.solve().items()}
    for k, (P, q, A, l, u) in enumerate(problems):
        w = [w for idx, w in g.groups if int(idx[0]) == k][0]
        ro = ob.OracleOSQP(P, q, A, l, u, perm=w.linsys().export_symbolic()["perm"], **kw).solve()
        assert relerr(r["x"][k].cpu().numpy(), ro["x"]) < 1e-8 and relerr(r["y"][k].cpu().numpy(), ro["y"]) < 1e-8, k
    # new values for every pattern in one update chain, then the same check on a few
    vals = []
    for idx, w in g.groups:
        P, q, A, l, u = problems[int(idx[0])]
        Pu = sparse.triu(sparse.csc_matrix(P), format="csc"); Pu.sort_indices()
        Ac = sparse.csc_matrix(A); Ac.sort_indices()
        vals.append((dev(Pu.data[None, :] * 1.03), dev(Ac.data[None, :] * 0.98)))
    g.update_P_A(vals)
    r2 = g.solve()
    for k in (0, 33, len(problems) - 1):
        P, q, A, l, u = problems[k]
        w = [w for idx, w in g.groups if int(idx[0]) == k][0]
        ro = ob.OracleOSQP(P * 1.03, q, A * 0.98, l, u, perm=w.linsys().export_symbolic()["perm"], **kw).solve()
        assert relerr(r2["x"][k].cpu().numpy(), ro["x"]) < 1e-8
    g.cleanup()


def test_pattern_groups_with_the_reference_defaults_run_in_one_chain(R):
    """Termination checks every 25 iterations, adaptive rho, scaling = 10 (include/constants.h:59-115: what a caller who changes nothing gets):
    the set's host loop is osqp_solve's loop for all groups at once -- groups of iterations up to the next check, one check launch, one
    masked refactorisation chain where rho moved, the active count of all groups read one check late.  Same kernels on the same data as
    the per-workspace route: bit-identical results, and status / iteration count / solution equal to the oracle per instance."""
    wls = [R.workloads.SharedPatternQPs(pattern_seed=s) for s in (2000, 2001, 2002, 2003, 2004)]
    problems = [wls[k % 5].instance(k // 5) for k in range(5 * 6 + 3)]
    for kw in (dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=4000, check_termination=25, adaptive_rho=1, adaptive_rho_interval=50, eps_abs=1e-4, eps_rel=1e-4,
                    warm_start=0, scaling=10),
               dict(rho=5.0, sigma=1e-6, alpha=1.6, max_iter=4000, check_termination=10, adaptive_rho=1, adaptive_rho_interval=30, eps_abs=1e-5, eps_rel=1e-5,
                    warm_start=0, scaling=0)):
        g1 = R.OSQPBatchGroups(problems, **kw)
        g0 = R.OSQPBatchGroups(problems, one_launch=False, **kw)
        if not any(os.environ.get(k) for k in ("RLDL_NO_TILE", "RLDL_NO_ARROW", "RLDL_CHECK_STAGED", "RLDL_NO_ARROW_FACTOR")):
            assert g1.one_launch and g1.groups_outside_the_chain == 0
        r1 = {k: v.clone() for k, v in g1.solve().items()}
        assert g1.one_launch or any(os.environ.get(k) for k in ("RLDL_NO_TILE", "RLDL_NO_ARROW", "RLDL_CHECK_STAGED", "RLDL_NO_ARROW_FACTOR"))   # (the solve did not drop the set)
        r0 = g0.solve()
        for key in ("x", "y", "z", "obj", "pri_res", "dua_res", "iter", "status"):
            assert torch.equal(r1[key], r0[key]), key
        assert int(r1["status"].min()) == 1 and len(set(r1["iter"].tolist())) > 1          # solved, after different numbers of iterations
        for k in (0, 7, 32):
            P, q, A, l, u = problems[k]
            w = [w for idx, w in g1.groups if int(k) in idx.tolist()][0]
            ro = ob.OracleOSQP(P, q, A, l, u, perm=w.linsys().export_symbolic()["perm"], **kw).solve()
            assert int(r1["iter"][k]) == ro["iter"] and int(r1["status"][k]) == ro["status"]
            assert relerr(r1["x"][k].cpu().numpy(), ro["x"]) < 1e-7 and relerr(r1["y"][k].cpu().numpy(), ro["y"]) < 1e-7
        g1.cleanup(); g0.cleanup()


def test_pattern_groups_values_from_the_producer_stream_and_changed_settings(R):
    """(1) update_P_A values that a torch kernel has just produced on torch's current stream: the chain runs on the set's own
    stream and must be ordered behind the producer (event wait) -- result bit-equal to the per-workspace route fed the same
    values after a full synchronisation.  (2) a settings change on a member workspace after creation takes the set off the single
    chain (osqp_multi_solve returns 2) instead of being ignored.  (3) a dest that is no permutation is refused by osqp_multi_create."""
    import ctypes as C
    from osqp_recursive_ldl_amd import _lib
    seeds = (2000, 2001, 2002)
    wls = [R.workloads.SharedPatternQPs(pattern_seed=s) for s in seeds]
    problems = [wls[k % 3].instance(k // 3) for k in range(3 * 40)]
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=30, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
    g1 = R.OSQPBatchGroups(problems, **kw)
    g0 = R.OSQPBatchGroups(problems, one_launch=False, **kw)
    base = []
    for idx, w in g1.groups:
        Pu = [sparse.triu(sparse.csc_matrix(problems[i][0]), format="csc") for i in idx.tolist()]
        Ac = [sparse.csc_matrix(problems[i][2]) for i in idx.tolist()]
        for M in Pu + Ac:
            M.sort_indices()
        base.append((dev(np.stack([M.data for M in Pu])), dev(np.stack([M.data for M in Ac]))))
    big = torch.ones((64, 1 << 20), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    for rep in range(3):
        fp, fa = 1.0 + 0.02 * (rep + 1), 1.0 - 0.01 * (rep + 1)
        big.mul_(1.0000001)                                         # keeps torch's stream busy in front of the producers
        vals = [(torch.mul(bp, fp), torch.mul(ba, fa)) for bp, ba in base]   # produced on the current stream, NOT synchronised
        g1.update_P_A(vals)
        del vals                                                    # the allocator may recycle the arrays only after the chain has read them
        junk = [torch.full_like(bp, float("nan")) for bp, _ in base]
        r1 = {k: v.clone() for k, v in g1.solve().items()}
        ref_vals = [(torch.mul(bp, fp), torch.mul(ba, fa)) for bp, ba in base]
        torch.cuda.synchronize()
        g0.update_P_A(ref_vals)
        r0 = g0.solve()
        for key in ("x", "y", "z", "obj", "iter", "status"):
            assert torch.equal(r1[key], r0[key]), (rep, key)
        del junk
    if g1.one_launch:
        # (2) max_iter changed on ONE member: the set no longer qualifies; the fallback honours every workspace's own settings
        assert g1.groups[1][1].update_settings(max_iter=10) == 0
        assert g0.groups[1][1].update_settings(max_iter=10) == 0
        r1, r0 = g1.solve(), g0.solve()
        assert not g1.one_launch
        for key in ("x", "y", "iter"):
            assert torch.equal(r1[key], r0[key]), key
        assert set(r1["iter"].tolist()) == {10, 30}
        # (3) dest validation
        hs = (C.c_void_p * 3)(*[w.h for _, w in g0.groups])
        for _, w in g0.groups:
            w.update_settings(max_iter=30)
        total = len(problems)
        for bad in (np.r_[np.arange(total - 1), total], np.r_[0, np.arange(total - 1)], np.r_[-1, np.arange(1, total)]):
            mh = C.c_void_p()
            d = np.ascontiguousarray(bad, dtype=np.int64)
            assert _lib.lib().osqp_multi_create(C.byref(mh), hs, 3, d.ctypes.data_as(_lib.IP), None) == 1 and not mh.value
    g1.cleanup(); g0.cleanup()


def test_update_settings_mirrors_the_reference_setters(R):
    """osqp_update_max_iter / _eps_abs / ... (src/osqp.c:1321-1560): range checks as in test_basic_qp.h:88-160, and the
    new values take effect (tighter eps -> more iterations, same count as a fresh workspace with those settings)."""
    d = load_golden("basic_qp")
    w = _solve_golden(R, d["P"], d["q"], d["A"], d["l"], d["u"], reps=2, eps_abs=1e-3, eps_rel=1e-3, check_termination=1, adaptive_rho=0)
    it0 = int(w.solve()["iter"][0])
    assert w.update_settings(max_iter=-1) == 1 and w.update_settings(eps_abs=-1.0) == 1 and w.update_settings(alpha=2.0) == 1
    assert w.update_settings(warm_start=2) == 1 and w.update_settings(check_termination=-1) == 1 and w.update_settings(eps_prim_inf=0.0) == 1
    assert w.update_settings(eps_abs=1e-7, eps_rel=1e-7, warm_start=0) == 0
    it1 = int(w.solve()["iter"][0])
    fresh = _solve_golden(R, d["P"], d["q"], d["A"], d["l"], d["u"], reps=2, eps_abs=1e-7, eps_rel=1e-7, check_termination=1, warm_start=0,
                          adaptive_rho=0)
    assert it1 == int(fresh.solve()["iter"][0]) > it0
    w.cleanup(); fresh.cleanup()


def test_enqueued_update_and_solve_equal_the_blocking_calls_and_report_a_failed_refactor_at_wait(R):
    """osqp_batch_update_P_A_async + osqp_batch_solve_async: same results as the blocking calls (same kernels, same
    inputs -> bit-identical); a refactorisation that fails (P made indefinite: fewer than n positive pivots,
    qdldl_interface.c:80-92) is reported by the next wait instead of by the update call."""
    wl = R.workloads.SharedPatternQPs()
    B = 6
    Px, Ax, q, l, u = wl.values(B)
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), **FIXED)
    Px2, Ax2 = wl.values(B, seed0=50)[:2]
    assert w.update_P_A(dev(Px2), dev(Ax2)) == 0
    ra = w.solve()
    assert w.update_P_A(dev(Px), dev(Ax)) == 0                 # back to the first problem, then the enqueued path
    assert w.update_P_A(dev(Px2), dev(Ax2), wait=False) == 0
    w.solve_async()
    rb = w.wait()
    assert torch.equal(ra["x"], rb["x"]) and torch.equal(ra["y"], rb["y"]) and torch.equal(ra["iter"], rb["iter"])
    bad = Px2.copy()
    bad[3] = -np.abs(bad[3]) - 1.0                              # instance 3: P negative definite
    assert w.update_P_A(dev(bad), dev(Ax2)) != 0                # blocking call: reported at once
    assert w.update_P_A(dev(bad), dev(Ax2), wait=False) == 0    # enqueued: reported by wait
    w.solve_async()
    with pytest.raises(RuntimeError):
        w.wait()
    assert w.update_P_A(dev(Px2), dev(Ax2), wait=False) == 0    # and the workspace is usable again
    w.solve_async()
    rc = w.wait()
    assert torch.equal(ra["x"], rc["x"])
    w.cleanup()


def test_update_bounds_refuses_crossed_bounds_and_refactor_failure_is_sticky(R):
    """osqp_update_bounds (src/osqp.c:805-813): l > u anywhere -> exitflag 1 and nothing changes.  And the verdict of an
    asynchronous refactorisation is sticky: a failing update_P_A enqueued BEFORE a good one is still reported by wait()."""
    wl = R.workloads.SharedPatternQPs(n=20, m=30, density=0.2, pattern_seed=7)
    B = 4
    Px, Ax, q, l, u = wl.values(B)
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=30, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), **kw)
    r0 = w.solve()["x"].cpu().numpy().copy()
    bad_l = l.copy(); bad_l[2, 5] = u[2, 5] + 1.0
    assert w.update_bounds(dev(bad_l), dev(u)) == 1
    assert np.array_equal(w.solve()["x"].cpu().numpy(), r0)     # bounds untouched
    assert w.update_bounds(dev(l), dev(u)) == 0
    Pbad = Px.copy()
    Pbad[1] = -Px[1]                                             # instance 1 becomes non-convex: fewer than n positive pivots
    assert w.update_P_A(dev(Pbad), dev(Ax), wait=False) == 0
    assert w.update_P_A(dev(Px), dev(Ax), wait=False) == 0       # a later, good refactorisation must not hide the failure
    with pytest.raises(RuntimeError):
        w.wait()
    assert w.update_P_A(dev(Px), dev(Ax), wait=False) == 0
    w.wait()                                                     # the verdict was cleared when it was read
    w.cleanup()


@pytest.mark.parametrize("scaling", [0, 10])
def test_enqueued_bounds_updates_keep_the_verdict_on_the_device(R, scaling):
    """osqp_batch_update_bounds_async / _partial_update_bounds_async: same result as the blocking calls; an update with l > u
    somewhere (osqp.c:805-813, recursive_ldl.c:137-145) changes nothing, is reported by the next wait() and does not stick."""
    wl = R.workloads.SharedPatternQPs(n=20, m=30, density=0.2, pattern_seed=7)
    B = 5
    Px, Ax, q, l, u = wl.values(B)
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=40, check_termination=0, adaptive_rho=0, warm_start=0, scaling=scaling)
    mk = lambda: R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), **kw)
    wa, wb = mk(), mk()
    l2, u2 = l - 0.3, u + 0.2
    assert wa.update_bounds(dev(l2), dev(u2)) == 0
    assert wb.update_bounds(dev(l2), dev(u2), wait=False) == 0
    ra = wa.solve(); wb.solve_async(); rb = wb.wait()
    assert torch.equal(ra["x"], rb["x"]) and torch.equal(ra["y"], rb["y"])
    lp, up = l[:, 4:11] - 0.7, u[:, 4:11] + 0.1                  # rows [4, 11)
    assert wa.partial_update_bounds(4, 11, dev(lp), dev(up)) == 0
    assert wb.partial_update_bounds(4, 11, dev(lp), dev(up), wait=False) == 0
    ra = wa.solve(); wb.solve_async(); rb = wb.wait()
    assert torch.equal(ra["x"], rb["x"]) and torch.equal(ra["y"], rb["y"])
    bad = lp.copy(); bad[3, 2] = up[3, 2] + 1.0                  # instance 3: l > u in one row
    assert wb.partial_update_bounds(4, 11, dev(bad), dev(up), wait=False) == 0      # enqueued: no verdict yet
    wb.solve_async()
    with pytest.raises(RuntimeError, match=r"\(1\)"):
        wb.wait()
    wb.solve_async()
    rc = wb.wait()                                               # verdict read and cleared; the bounds are those of before
    assert torch.equal(rc["x"], rb["x"])
    assert wb.update_bounds(dev(l), dev(u), wait=False) == 0     # and a later good update goes through
    wb.solve_async(); rd = wb.wait()
    assert wa.update_bounds(dev(l), dev(u)) == 0
    assert torch.equal(rd["x"], wa.solve()["x"])
    wa.cleanup(); wb.cleanup()

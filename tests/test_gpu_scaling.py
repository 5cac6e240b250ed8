"""GPU parity tests for the device-side Ruiz equilibration (src/scaling.c) and everything downstream of it in the
batched ADMM driver: termination on unscaled residuals (auxil.c:243-362), unscaled infeasibility certificates
(auxil.c:364-515, :757-775), unscale_solution (scaling.c:175-192), and the scaling-aware update entry points
(osqp.c:770-774, :822-826, :937-942, :1183-1241).  The checker is the CPU oracle run with the same settings and the
same KKT permutation; the reference's own basic_qp / update_matrices fixtures (generated with its default
scaling = 10) pin the end results.  Tolerances: fp64, 1e-8 relative on iterates after a fixed number of iterations,
TESTS_TOL (1e-4, the reference's own test tolerance) against fixtures."""
import numpy as np
import pytest

from scipy import sparse

import oracle_bindings as ob
from helpers import load_golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

TESTS_TOL = 1e-4
OSQP_INFTY = 1e30
OSQP_NAN = float(0x7fc00000)        # include/constants.h:112 -- a large finite number in the reference, not an IEEE NaN


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to("cuda:0")


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1.0, float(np.max(np.abs(b)))))


def osqp_inf(v):
    v = np.asarray(v, float).copy()
    v[v > OSQP_INFTY] = OSQP_INFTY; v[v < -OSQP_INFTY] = -OSQP_INFTY
    return v


@pytest.fixture(scope="module")
def R():
    import osqp_recursive_ldl_amd as R
    return R


def test_ruiz_scaling_vectors_match_oracle(R):
    wl = R.workloads.SharedPatternQPs()
    B = 5
    Px, Ax, q, l, u = wl.values(B)
    Px[1] *= 1e3; Ax[2] *= 1e-3; q[3] *= 1e5          # badly scaled instances exercise limit_scaling and c
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=10, check_termination=0, adaptive_rho=0, warm_start=0, scaling=10)
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), **kw)
    assert w.status == 0
    D, E, c = w.scaling_vectors()
    perm = w.linsys().export_symbolic()["perm"]
    for b in range(B):
        P, qq, A, ll, uu = wl.instance(b)
        P = P.copy(); A = A.copy()
        P.data[:] = Px[b]; A.data[:] = Ax[b]
        o = ob.OracleOSQP(P, q[b], A, ll, uu, perm=perm, **kw)
        Do, Eo, co = o.scaling_vectors()
        assert relerr(D[b].cpu().numpy(), Do) < 1e-12
        assert relerr(E[b].cpu().numpy(), Eo) < 1e-12
        assert abs(float(c[b]) - co) < 1e-12 * max(1.0, abs(co))
        # the factor of the equilibrated KKT matrix
        f = w.linsys().export_factor(b)
        assert relerr(f["Lx"], o.linsys_export()["Lx"]) < 1e-10
        o.cleanup()
    w.cleanup()


def test_scaled_admm_fixed_iterations_match_oracle(R):
    wl = R.workloads.SharedPatternQPs()
    B = 4
    Px, Ax, q, l, u = wl.values(B)
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=150, check_termination=0, adaptive_rho=0, warm_start=0, scaling=10)
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), **kw)
    r = w.solve()
    perm = w.linsys().export_symbolic()["perm"]
    for b in range(B):
        P, qq, A, ll, uu = wl.instance(b)
        ro = ob.OracleOSQP(P, qq, A, ll, uu, perm=perm, **kw).solve()
        assert relerr(r["x_iter"][b].cpu().numpy(), ro["x_iter"]) < 1e-8
        assert relerr(r["y_iter"][b].cpu().numpy(), ro["y_iter"]) < 1e-8
        assert relerr(r["z"][b].cpu().numpy(), ro["z_iter"]) < 1e-8
        assert relerr(r["x"][b].cpu().numpy(), ro["x"]) < 1e-8         # unscaled solution
        assert relerr(r["y"][b].cpu().numpy(), ro["y"]) < 1e-8
        assert int(r["status"][b]) == ro["status"] and int(r["iter"][b]) == ro["iter"] == 150
        assert abs(float(r["pri_res"][b]) - ro["pri_res"]) < 1e-8 * max(1, ro["pri_res"])
        assert abs(float(r["dua_res"][b]) - ro["dua_res"]) < 1e-8 * max(1, ro["dua_res"])
        assert abs(float(r["obj"][b]) - ro["obj"]) < 1e-8 * max(1, abs(ro["obj"]))
    w.cleanup()


@pytest.mark.parametrize("scaled_termination", [0, 1])
def test_scaled_termination_and_adaptive_rho_match_oracle(R, scaled_termination):
    wl = R.workloads.SharedPatternQPs(n=20, m=30, density=0.25, pattern_seed=21)
    B = 6
    Px, Ax, q, l, u = wl.values(B)
    l[:, :4] = u[:, :4] = 0.5 * (l[:, :4] + u[:, :4])
    u[:, 4] = 1e30; l[:, 5] = -1e30; l[:, 6] = -1e30; u[:, 6] = 1e30
    Ax[:, ::3] *= 30.0
    kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=4000, check_termination=25, adaptive_rho=1, adaptive_rho_interval=50,
              eps_abs=1e-5, eps_rel=1e-5, warm_start=0, scaling=10, scaled_termination=scaled_termination)
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), **kw)
    r = w.solve()
    perm = w.linsys().export_symbolic()["perm"]
    for b in range(B):
        P, qq, A, ll, uu = wl.instance(b)
        P = P.copy(); A = A.copy(); P.data[:] = Px[b]; A.data[:] = Ax[b]
        ro = ob.OracleOSQP(P, q[b], A, l[b], u[b], perm=perm, **kw).solve()
        assert int(r["status"][b]) == ro["status"] == 1
        assert int(r["iter"][b]) == ro["iter"], (b, int(r["iter"][b]), ro["iter"])
        assert relerr(r["x"][b].cpu().numpy(), ro["x"]) < 1e-7
        assert relerr(r["y"][b].cpu().numpy(), ro["y"]) < 1e-7
        assert abs(float(r["obj"][b]) - ro["obj"]) < 1e-7 * max(1, abs(ro["obj"]))
    w.cleanup()


def _triu(P):
    M = sparse.triu(sparse.csc_matrix(P), format="csc"); M.sort_indices()
    return M


def _csc(A):
    M = sparse.csc_matrix(A); M.sort_indices()
    return M


def _golden_batch(R, P, q, A, l, u, reps=2, **kw):
    Pc, Ac = R.CscPattern(P, upper=True), R.CscPattern(A)
    base = dict(eps_abs=1e-7, eps_rel=1e-7, max_iter=20000, check_termination=1, scaling=10, adaptive_rho=1, adaptive_rho_interval=25)
    base.update(kw)
    tile = lambda v: dev(np.tile(np.asarray(v, float), (reps, 1)))
    return R.OSQPBatch(Pc, Ac, tile(Pc.x), tile(Ac.x), tile(q), tile(osqp_inf(l)), tile(osqp_inf(u)), **base), Pc, Ac


def test_basic_qp_fixture_with_reference_default_scaling(R):
    d = load_golden("basic_qp"); s = d["sols"]
    w, Pc, Ac = _golden_batch(R, d["P"], d["q"], d["A"], d["l"], d["u"])
    r = w.solve()
    for b in range(2):
        assert int(r["status"][b]) == 1
        assert np.max(np.abs(r["x"][b].cpu().numpy() - s["x_test"])) < TESTS_TOL
        assert np.max(np.abs(r["y"][b].cpu().numpy() - s["y_test"])) < TESTS_TOL
        assert abs(float(r["obj"][b]) - s["obj_value_test"]) < TESTS_TOL
    # update_lin_cost / update_bounds / warm_start on equilibrated data vs a fresh oracle with the same updates
    perm = w.linsys().export_symbolic()["perm"]
    kw = dict(eps_abs=1e-7, eps_rel=1e-7, max_iter=20000, check_termination=1, scaling=10, adaptive_rho=1, adaptive_rho_interval=25)
    o = ob.OracleOSQP(_triu(d["P"]), d["q"], _csc(d["A"]), osqp_inf(d["l"]), osqp_inf(d["u"]), perm=perm, **kw)
    o.solve()
    qn = np.asarray(s["q_new"], float); ln = osqp_inf(s["l_new"]); un = osqp_inf(s["u_new"])
    tile = lambda v: dev(np.tile(np.asarray(v, float), (2, 1)))
    assert w.update_lin_cost(tile(qn)) == 0 and w.update_bounds(tile(ln), tile(un)) == 0
    o.update_lin_cost(qn); o.update_bounds(ln, un)
    r = w.solve(); ro = o.solve()
    for b in range(2):
        assert int(r["status"][b]) == ro["status"] == 1 and int(r["iter"][b]) == ro["iter"]
        assert relerr(r["x"][b].cpu().numpy(), ro["x"]) < 1e-7
        assert relerr(r["y"][b].cpu().numpy(), ro["y"]) < 1e-7
    x0 = np.asarray(ro["x"]) * 0.9; y0 = np.asarray(ro["y"]) * 1.1
    assert w.warm_start(tile(x0), tile(y0)) == 0
    o.warm_start(x0, y0)
    r = w.solve(); ro = o.solve()
    for b in range(2):
        assert int(r["iter"][b]) == ro["iter"] and int(r["status"][b]) == ro["status"]
        assert relerr(r["x"][b].cpu().numpy(), ro["x"]) < 1e-7
    w.cleanup(); o.cleanup()


def test_update_matrices_fixture_with_scaling(R):
    """test_update_matrices.h:112-227 (osqp_update_P / _A / _P_A with the reference's default scaling)."""
    d = load_golden("update_matrices")["data"]
    P, Pn, A, An = d["test_solve_Pu"], d["test_solve_Pu_new"], d["test_solve_A"], d["test_solve_A_new"]
    Pc, Ac = R.CscPattern(P, upper=True), R.CscPattern(A)
    Pnc, Anc = R.CscPattern(Pn, upper=True), R.CscPattern(An)
    assert np.array_equal(Pc.i, Pnc.i) and np.array_equal(Ac.i, Anc.i)
    kw = dict(eps_abs=1e-7, eps_rel=1e-7, max_iter=20000, check_termination=1, scaling=10, adaptive_rho=1, adaptive_rho_interval=25)
    tile = lambda v: dev(np.tile(np.asarray(v, float), (2, 1)))
    w = R.OSQPBatch(Pc, Ac, tile(Pc.x), tile(Ac.x), tile(d["test_solve_q"]), tile(osqp_inf(d["test_solve_l"])),
                    tile(osqp_inf(d["test_solve_u"])), **kw)
    r = w.solve()
    assert np.max(np.abs(r["x"][0].cpu().numpy() - d["test_solve_x"])) < TESTS_TOL
    assert np.max(np.abs(r["y"][0].cpu().numpy() - d["test_solve_y"])) < TESTS_TOL
    assert w.update_P_A(tile(Pnc.x), None) == 0
    r = w.solve()
    assert np.max(np.abs(r["x"][1].cpu().numpy() - d["test_solve_P_new_x"])) < TESTS_TOL
    assert np.max(np.abs(r["y"][1].cpu().numpy() - d["test_solve_P_new_y"])) < TESTS_TOL
    assert w.update_P_A(tile(Pnc.x), tile(Anc.x)) == 0
    r = w.solve()
    assert np.max(np.abs(r["x"][0].cpu().numpy() - d["test_solve_P_A_new_x"])) < TESTS_TOL
    assert np.max(np.abs(r["y"][0].cpu().numpy() - d["test_solve_P_A_new_y"])) < TESTS_TOL
    # and step-for-step against the oracle doing the same unscale -> write -> scale sequence
    perm = w.linsys().export_symbolic()["perm"]
    o = ob.OracleOSQP(_triu(P), d["test_solve_q"], _csc(A), osqp_inf(d["test_solve_l"]), osqp_inf(d["test_solve_u"]), perm=perm, **kw)
    o.solve(); o.update_P_A(Pnc.x, None); o.solve(); o.update_P_A(Pnc.x, Anc.x); ro = o.solve()
    assert int(r["iter"][0]) == ro["iter"] and int(r["status"][0]) == ro["status"]
    assert relerr(r["x"][0].cpu().numpy(), ro["x"]) < 1e-7
    D, E, c = w.scaling_vectors(); Do, Eo, co = o.scaling_vectors()
    assert relerr(D[0].cpu().numpy(), Do) < 1e-12 and relerr(E[0].cpu().numpy(), Eo) < 1e-12 and abs(float(c[0]) - co) < 1e-12 * max(1, co)
    w.cleanup(); o.cleanup()


def test_infeasibility_certificates_with_scaling(R):
    """test_primal_dual_infeasibility.h with scaling on: statuses, OSQP_NAN solution, normalised unscaled certificates."""
    g = load_golden("primal_dual_infeasibility")["data"]
    kw = dict(max_iter=2000, eps_abs=1e-6, eps_rel=1e-6, check_termination=1, scaling=10, adaptive_rho=1, adaptive_rho_interval=25,
              warm_start=0)
    tile = lambda v: dev(np.tile(np.asarray(v, float), (2, 1)))
    seen = set()
    for Ak, uk in (("A12", "u1"), ("A12", "u2"), ("A34", "u3"), ("A34", "u4")):
        Pc, Ac = R.CscPattern(g["P"], upper=True), R.CscPattern(g[Ak])
        l, u, q = osqp_inf(g["l"]), osqp_inf(g[uk]), np.asarray(g["q"], float)
        w = R.OSQPBatch(Pc, Ac, tile(Pc.x), tile(Ac.x), tile(q), tile(l), tile(u), **kw)
        r = w.solve()
        perm = w.linsys().export_symbolic()["perm"]
        ro = ob.OracleOSQP(_triu(g["P"]), q, _csc(g[Ak]), l, u, perm=perm, **kw).solve()
        assert int(r["status"][0]) == ro["status"], (Ak, uk)
        assert int(r["iter"][0]) == ro["iter"], (Ak, uk)
        seen.add(ro["status"])
        if ro["status"] in (-3, 3):
            assert bool((r["x"][0] == OSQP_NAN).all()) and (ro["x"] == OSQP_NAN).all()   # vec_set_scalar(x, OSQP_NAN)
            assert relerr(r["delta_y"][0].cpu().numpy(), ro["delta_y"]) < 1e-7
            assert float(r["obj"][0]) == OSQP_INFTY
        elif ro["status"] in (-4, 4):
            assert bool((r["x"][0] == OSQP_NAN).all()) and bool((r["y"][0] == OSQP_NAN).all())
            assert relerr(r["delta_x"][0].cpu().numpy(), ro["delta_x"]) < 1e-7
            assert float(r["obj"][0]) == -OSQP_INFTY
        else:
            assert np.max(np.abs(r["x"][0].cpu().numpy() - g["x1"])) < TESTS_TOL
            assert np.max(np.abs(r["y"][0].cpu().numpy() - g["y1"])) < TESTS_TOL
            assert relerr(r["x"][0].cpu().numpy(), ro["x"]) < 1e-7
        w.cleanup()
    assert {1, -3, -4} <= seen


@pytest.mark.parametrize("scaling", [0, 10])
def test_polish_matches_oracle_and_fixture(R, scaling):
    """Batched polish (src/polish.c) on the shared pattern: status_polish, polished x / y / obj / residuals vs the oracle's
    reduced-matrix polish, and the basic_qp optimum to 1e-9 from eps = 1e-3 ADMM iterates."""
    d = load_golden("basic_qp"); s = d["sols"]
    kw = dict(eps_abs=1e-3, eps_rel=1e-3, max_iter=4000, check_termination=25, scaling=scaling, adaptive_rho=1, adaptive_rho_interval=100,
              polish=1, polish_refine_iter=3, delta=1e-6)
    w, Pc, Ac = _golden_batch(R, d["P"], d["q"], d["A"], d["l"], d["u"], **kw)
    r = w.solve()
    assert int(r["status"][0]) == 1 and int(r["status_polish"][0]) == 1
    assert np.max(np.abs(r["x"][0].cpu().numpy() - s["x_test"])) < 1e-9
    assert np.max(np.abs(r["y"][0].cpu().numpy() - s["y_test"])) < 1e-9
    assert abs(float(r["obj"][0]) - s["obj_value_test"]) < 1e-9
    w.cleanup()
    # metric shape: per-instance comparison with the oracle (which removes the inactive rows instead of zeroing them)
    wl = R.workloads.SharedPatternQPs()
    B = 6
    Px, Ax, q, l, u = wl.values(B)
    w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), dev(q), dev(l), dev(u), **kw)
    r = w.solve()
    perm = w.linsys().export_symbolic()["perm"]
    for b in range(B):
        P, qq, A, ll, uu = wl.instance(b)
        ro = ob.OracleOSQP(P, qq, A, ll, uu, perm=perm, **kw).solve()
        assert int(r["status"][b]) == ro["status"] and int(r["iter"][b]) == ro["iter"]
        assert int(r["status_polish"][b]) == ro["status_polish"]
        assert relerr(r["x"][b].cpu().numpy(), ro["x"]) < 1e-8 and relerr(r["y"][b].cpu().numpy(), ro["y"]) < 1e-8
        assert abs(float(r["obj"][b]) - ro["obj"]) < 1e-8 * max(1, abs(ro["obj"]))
        if ro["status_polish"] == 1:
            assert float(r["pri_res"][b]) < 1e-9 and float(r["dua_res"][b]) < 1e-9
    w.cleanup()

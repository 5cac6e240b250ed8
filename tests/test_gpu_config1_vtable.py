"""BASELINE config 1: tests/basic_qp through the LEGACY VTABLE.  The CPU oracle runs the reference's ADMM loop
(osqp_solve, src/osqp.c:354-519) and calls the plugin object exactly where the reference does:
  work->linsys_solver->solve(work->linsys_solver, work->xz_tilde)            src/auxil.c:180-186
  work->linsys_solver->update_rho_vec(work->linsys_solver, work->rho_vec)    src/osqp.c:1310-1318
with `linsys_solver` = the object built by init_linsys_solver_hipldl (struct prefix of include/types.h:298-319), i.e. the
function pointers of the HIP library.  The iterates must reach the reference's known optimum (tests/basic_qp/test_basic_qp.h,
tests/basic_qp2/test_basic_qp2.h, TESTS_TOL 1e-4) with the iteration count of the oracle's own backend.
Also here: the primal_infeasibility fixture (n = 50, m = 150) on the batched driver."""
import ctypes as C

import numpy as np
import pytest

import oracle_bindings as ob
from helpers import load_golden, osqp_inf

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
TESTS_TOL = 1e-4


def run_through_vtable(d, **kw):
    import osqp_recursive_ldl_amd as R
    args = (d["P"], d["q"], d["A"], osqp_inf(d["l"]), osqp_inf(d["u"]))
    own = ob.OracleOSQP(*args, **kw)
    r_own = own.solve()
    w = ob.OracleOSQP(*args, **kw)
    P, A, sigma, rho_vec = w.backend_data()                      # what osqp_setup hands to init_linsys_solver (osqp.c:157-160)
    plug = R.HipLDLSolver(R.CscPattern(P, upper=True), R.CscPattern(A), sigma, rho_vec)
    assert plug.status == 0 and plug.type is not None
    vt = plug._sp.contents
    calls = {"solve": 0, "rho": 0}
    SOLVE = C.CFUNCTYPE(C.c_longlong, C.c_void_p, C.POINTER(C.c_double))
    RHO = C.CFUNCTYPE(C.c_longlong, C.c_void_p, C.POINTER(C.c_double))
    raw_solve = C.cast(vt.solve, C.c_void_p).value
    raw_rho = C.cast(vt.update_rho_vec, C.c_void_p).value

    def solve(self_p, b):                                        # thin counters around the plugin's own function pointers
        calls["solve"] += 1
        return SOLVE(raw_solve)(self_p, b)

    def rho(self_p, rv):
        calls["rho"] += 1
        return RHO(raw_rho)(self_p, rv)
    cs, cr = SOLVE(solve), RHO(rho)
    w.use_external_linsys(plug._sp, cs, cr)
    r = w.solve()
    assert calls["solve"] == r["iter"]                           # one ->solve per ADMM iteration (auxil.c:185)
    assert calls["rho"] == r["rho_updates"]                      # one ->update_rho_vec per accepted rho change
    plug.free()
    return r, r_own


def test_basic_qp_through_the_legacy_vtable():
    d = load_golden("basic_qp"); s = d["sols"]
    kw = dict(eps_abs=1e-7, eps_rel=1e-7, max_iter=20000, check_termination=1, scaling=10, adaptive_rho=1, adaptive_rho_interval=25)
    r, r_own = run_through_vtable(d, **kw)
    assert r["status"] == 1 and r["iter"] == r_own["iter"] and r["rho_updates"] == r_own["rho_updates"] > 0
    assert np.max(np.abs(r["x"] - s["x_test"])) < TESTS_TOL
    assert np.max(np.abs(r["y"] - s["y_test"])) < TESTS_TOL
    assert abs(r["obj"] - s["obj_value_test"]) < TESTS_TOL
    assert np.max(np.abs(r["x"] - r_own["x"])) < 1e-9


def test_basic_qp2_through_the_legacy_vtable():
    d = load_golden("basic_qp2"); s = d["sols"]
    kw = dict(eps_abs=1e-9, eps_rel=1e-9, max_iter=200000, check_termination=1, scaling=10, adaptive_rho=1, adaptive_rho_interval=25)
    r, r_own = run_through_vtable(d, **kw)
    assert r["status"] == 1 and r["iter"] == r_own["iter"]
    assert np.max(np.abs(r["x"] - s["x_test"])) < TESTS_TOL * 10
    assert abs(r["obj"] - s["obj_value_test"]) / abs(s["obj_value_test"]) < TESTS_TOL


def test_primal_infeasibility_fixture_on_the_batched_driver():
    """tests/primal_infeasibility/test_primal_infeasibility.h:27-52: OSQP_PRIMAL_INFEASIBLE, same iteration count as the oracle."""
    import osqp_recursive_ldl_amd as R
    d = load_golden("primal_infeasibility")
    kw = dict(max_iter=10000, alpha=1.6, scaling=0, warm_start=0, check_termination=25, adaptive_rho=1, adaptive_rho_interval=100,
              eps_abs=1e-3, eps_rel=1e-3)
    ro = ob.OracleOSQP(d["P"], d["q"], d["A"], osqp_inf(d["l"]), osqp_inf(d["u"]), **kw).solve()
    assert ro["status"] == -3
    Pc, Ac = R.CscPattern(d["P"], upper=True), R.CscPattern(d["A"])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to("cuda:0")
    tile = lambda v: dev(np.tile(np.asarray(v, float), (2, 1)))
    w = R.OSQPBatch(Pc, Ac, tile(Pc.x), tile(Ac.x), tile(d["q"]), tile(osqp_inf(d["l"])), tile(osqp_inf(d["u"])), **kw)
    assert w.status == 0
    r = w.solve()
    for b in range(2):
        assert int(r["status"][b]) == -3 and int(r["iter"][b]) == ro["iter"]
        assert float(r["obj"][b]) == 1e30                       # OSQP_INFTY (osqp.c:565-569)
    w.cleanup()

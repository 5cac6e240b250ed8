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

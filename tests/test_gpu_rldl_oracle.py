"""k_stage_factor_r (the HIP stage recursion: full factorisation and restart at a stage) against the oracle's restatement of
the reference's recursion (oracle/rldl_oracle.c <- src/recursive_ldl.c:554-1318), block by block: the closed-form
permutation (integers, exact), L in the reference's emission order, D with the constraint blocks' pivots negative.
The oracle runs with the drop thresholds off and the terminal rows' own rho (the HIP path keeps the fixed dense-block
pattern and factorises the assembled KKT matrix); a second oracle run with the reference's drops on shows that what the
reference would leave out is below its thresholds.  "Parity unpinned by the reference": no fixture exists for this path."""
import numpy as np
import pytest
from scipy import sparse

import oracle_bindings as ob

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
SIGMA = 1e-6


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to("cuda:0")


def rho_of(wl, l, u, rho=0.1):
    return np.where(np.abs(u - l) < 1e-4, 1e3 * rho, rho) * (1.0 + 0.1 * np.arange(wl.m) / wl.m)


def gpu_L(sym, f, n):
    return sparse.csc_matrix((f["Lx"], sym["Li"], sym["Lp"]), shape=(n, n)).toarray()


@pytest.mark.parametrize("N", [1, 2, 7, 20])
def test_stage_factor_matches_the_stage_recursion_oracle(N):
    import osqp_recursive_ldl_amd as R
    wl = R.workloads.MPCStageQPs(N=N)
    B = 3
    Px, Ax, q, l, u = wl.values(B)
    rho = np.stack([rho_of(wl, l[b], u[b]) for b in range(B)])
    ls = R.BatchLinsys.recursive(wl.dims, wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), SIGMA, dev(rho))
    assert ls.status == 0
    sym = ls.export_symbolic()
    Nk = wl.n + wl.m
    for b in range(B):
        P, qq, A, ll, uu = wl.instance(b)
        o = ob.OracleRLDL(wl.dims)
        assert o.factor(P, A, SIGMA, 1.0 / rho[b]) > 0
        assert np.array_equal(sym["perm"], o.perm)               # integers: bit-exact
        f = ls.export_factor(b)
        Lo = o.L().toarray()
        Lg = gpu_L(sym, f, Nk)
        # outside the symbolic pattern the recursion leaves only rounding noise: it forms L21 as Ybar' (L + I), a product whose
        # terms cancel where L21 is structurally zero (cs_addon.c:274-339)
        assert np.all(np.abs(Lo[Lg == 0]) <= 1e-12 * max(1.0, np.max(np.abs(Lo))))
        assert np.max(np.abs(Lg - Lo)) <= 1e-11 * max(1.0, np.max(np.abs(Lo)))
        assert np.max(np.abs(f["Dinv"] - o.Dinv) / np.abs(o.Dinv)) <= 1e-11
        assert np.array_equal(np.sign(f["D"]), np.sign(o.Dinv))  # constraint blocks: negative pivots (Dinv = -Dinv, :673-675)
        thin = ob.OracleRLDL(wl.dims, mirror_drops=1)
        assert thin.factor(P, A, SIGMA, 1.0 / rho[b]) > 0
        Lt = thin.L().toarray()
        assert np.all(np.abs(Lg[(Lt == 0) & (Lg != 0)]) <= 1e-8)                # entries the reference would not store are tiny
    ls.free()


@pytest.mark.parametrize("N,k", [(2, 1), (7, 3), (7, 6), (20, 19)])
def test_restart_at_a_stage_matches_update_from_pivot(N, k):
    """rldl_batch_update_from_stage(k): data of stages >= k changed.  Reference: LDL_update_from_pivot resumes at the cached
    constraint block of stage k - 1 (:969-970, :995-997); both must give the factor of the new matrix, and the HIP path must
    leave the kept columns untouched."""
    import osqp_recursive_ldl_amd as R
    wl = R.workloads.MPCStageQPs(N=N)
    B = 2
    Px, Ax, q, l, u = wl.values(B)
    rho = np.stack([rho_of(wl, l[b], u[b]) for b in range(B)])
    ls = R.BatchLinsys.recursive(wl.dims, wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), SIGMA, dev(rho))
    sym = ls.export_symbolic()
    Nk = wl.n + wl.m
    before = [ls.export_factor(b)["Lx"].copy() for b in range(B)]
    rg = np.random.default_rng(11)
    Px2 = Px * np.where(wl.P_stage >= k, 1 + 0.05 * rg.standard_normal(Px.shape), 1.0)
    Ax2 = Ax * np.where((wl.A_stage >= k) & (Ax != -1.0), 1 + 0.05 * rg.standard_normal(Ax.shape), 1.0)
    assert ls.update_from_stage(k, dev(Px2), dev(Ax2), None) == 0
    nx, nu, ny = wl.dims[1], wl.dims[2], wl.dims[3]
    keep_cols = nu + (k - 1) * (2 * nx + nu + ny)               # Q_0, C_0, ..., Q_{k-1}: what the reference keeps (:969)
    for b in range(B):
        P0, _, A0, _, _ = wl.instance(b)
        P2 = sparse.csc_matrix((Px2[b], wl.P_pattern.indices, wl.P_pattern.indptr), shape=P0.shape)
        A2 = sparse.csc_matrix((Ax2[b], wl.A_pattern.indices, wl.A_pattern.indptr), shape=A0.shape)
        o = ob.OracleRLDL(wl.dims)
        assert o.factor(P0, A0, SIGMA, 1.0 / rho[b]) > 0
        assert o.factor(P2, A2, SIGMA, 1.0 / rho[b], iter_start=k - 1) > 0
        f = ls.export_factor(b)
        Lg, Lo = gpu_L(sym, f, Nk), o.L().toarray()
        assert np.max(np.abs(Lg - Lo)) <= 1e-11 * max(1.0, np.max(np.abs(Lo)))
        assert np.max(np.abs(f["Dinv"] - o.Dinv) / np.abs(o.Dinv)) <= 1e-11
        kept = sym["Lp"][keep_cols]
        assert np.array_equal(f["Lx"][:kept], before[b][:kept])  # columns of the kept blocks: bit-identical
    ls.free()

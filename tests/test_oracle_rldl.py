"""The oracle's restatement of the reference's stage recursion (oracle/rldl_oracle.c: pivot_even / pivot_odd / pivot_final,
LDL_factorize_recursive, LDL_update_from_pivot of src/recursive_ldl.c:554-1318 with the cs_addon.c products) -- CPU only.

The reference holds no fixture for this path and cannot be built here: "parity unpinned by the reference".  What these tests
pin is that the restatement (block order, negated Dinv / L21 of the constraint pivots, the closed-form permutation, the
X_even[] restart) computes the LDL' factor of the assembled, stage-permuted KKT matrix: P K P' = L D L' and equality with the
generic oracle's factor of the same permuted matrix; and that the mirrored quirks (drop thresholds, terminal-rho index) do
exactly what the header of rldl_oracle.c says."""
import numpy as np
import pytest
from scipy import sparse

import oracle_bindings as ob
import osqp_recursive_ldl_amd as R

SIGMA = 1e-6


def problem(N, b=0, rho=0.1):
    wl = R.workloads.MPCStageQPs(N=N)
    P, q, A, l, u = wl.instance(b)
    rv = np.where(np.abs(u - l) < 1e-4, 1e3 * rho, rho) * (1.0 + 0.1 * np.arange(wl.m) / wl.m)   # rows differ: an index slip shows
    return wl, P, A, 1.0 / rv


def kkt(P, A, rho_inv):
    Pf = P + P.T - sparse.diags(P.diagonal())
    return sparse.bmat([[Pf + SIGMA * sparse.eye(P.shape[0]), A.T], [A, -sparse.diags(rho_inv)]], format="csc").toarray()


def ldl_residual(o, K):
    L = o.L().toarray() + np.eye(K.shape[0])
    D = np.diag(1.0 / o.Dinv)
    Kp = K[np.ix_(o.perm, o.perm)]
    return np.max(np.abs(L @ D @ L.T - Kp)) / np.max(np.abs(K))


@pytest.mark.parametrize("N", [1, 2, 7, 20])
def test_stage_recursion_factors_the_permuted_kkt_and_equals_the_generic_oracle(N):
    wl, P, A, ri = problem(N)
    o = ob.OracleRLDL(wl.dims)
    assert o.factor(P, A, SIGMA, ri) > 0
    assert np.array_equal(o.perm, R.workloads.stage_permutation(*wl.dims))          # compute_permutations :1345-1363
    assert ldl_residual(o, kkt(P, A, ri)) < 1e-13
    # signs: cost blocks positive pivots, constraint blocks negative ones (Dinv = -Dinv, :673-675, :931)
    isvar = o.perm < wl.n
    assert np.all(o.Dinv[isvar] > 0) and np.all(o.Dinv[~isvar] < 0)
    g = ob.OracleLinsys(P, A, SIGMA, 1.0 / ri, perm=o.perm)
    assert g.status == 0
    e = g.export()
    Lg = sparse.csc_matrix((e["Lx"], e["Li"], e["Lp"]), shape=(wl.n + wl.m,) * 2).toarray()
    assert np.max(np.abs(o.L().toarray() - Lg)) <= 1e-12 * max(1.0, np.max(np.abs(Lg)))
    assert np.max(np.abs(o.Dinv - e["Dinv"]) / np.abs(e["Dinv"])) <= 1e-12


def test_drop_thresholds_thin_the_pattern_but_not_the_factor():
    wl, P, A, ri = problem(7)
    full, thin = ob.OracleRLDL(wl.dims, mirror_drops=0), ob.OracleRLDL(wl.dims, mirror_drops=1)
    nf, nt = full.factor(P, A, SIGMA, ri), thin.factor(P, A, SIGMA, ri)
    assert 0 < nt <= nf
    Lf, Lt = full.L().toarray(), thin.L().toarray()
    dropped = (Lt == 0) & (Lf != 0)
    assert np.all(np.abs(Lf[dropped]) <= 1e-9)                  # what is missing was below the thresholds (1e-10 on L, 1e-9 through Ybar')
    # the thinned Ybar' (entries up to 1e-9 gone) feeds the next blocks, where the stiff rho of the dynamics rows (1e2) and the
    # small pivots of the input costs magnify it: the reference's factor is exact only to about 1e-6 on this problem
    assert np.max(np.abs(Lf - Lt)) <= 1e-5
    assert 1e-14 < ldl_residual(thin, kkt(P, A, ri)) < 1e-6


def test_terminal_block_reads_rho_of_the_last_interior_row_block():
    """recursive_ldl.c:1085, :1293: rho_inv_vec[(Nmax - 1) (nx + ny)] for the terminal rows."""
    wl, P, A, ri = problem(5)
    N, nx, nu, ny, nt = wl.dims
    o = ob.OracleRLDL(wl.dims, terminal_rho_own=0)
    assert o.factor(P, A, SIGMA, ri) > 0
    quirk = ri.copy()
    quirk[N * (nx + ny):] = ri[(N - 1) * (nx + ny):(N - 1) * (nx + ny) + nt]
    assert ldl_residual(o, kkt(P, A, quirk)) < 1e-13
    assert ldl_residual(o, kkt(P, A, ri)) > 1e-6                # ... and not the matrix with the terminal rows' own rho


@pytest.mark.parametrize("N,s", [(2, 0), (7, 3), (7, 5), (20, 18)])
def test_restart_from_the_cached_block_equals_a_full_factorisation(N, s):
    """LDL_update_from_pivot (:946-1110): data of stages > s changed -> resume at the constraint block of stage s with
    X_even[s]; the kept columns and the recomputed ones together are the factor of the new matrix, bit for bit."""
    wl, P, A, ri = problem(N)
    o = ob.OracleRLDL(wl.dims)
    assert o.factor(P, A, SIGMA, ri) > 0
    rng = np.random.default_rng(3)
    P2, A2 = P.copy(), A.copy()
    pm = wl.P_stage > s
    am = (wl.A_stage > s) & (A.data != -1.0)
    P2.data[pm] *= 1 + 0.05 * rng.standard_normal(int(pm.sum()))
    A2.data[am] *= 1 + 0.05 * rng.standard_normal(int(am.sum()))
    keep_cols = wl.dims[2] + s * (2 * wl.dims[1] + wl.dims[2] + wl.dims[3])
    before = (o.Lp[:keep_cols + 1].copy(), o.Lx[:o.Lp[keep_cols]].copy())
    assert o.factor(P2, A2, SIGMA, ri, iter_start=s) > 0
    assert np.array_equal(before[0], o.Lp[:keep_cols + 1]) and np.array_equal(before[1], o.Lx[:o.Lp[keep_cols]])
    ref = ob.OracleRLDL(wl.dims)
    assert ref.factor(P2, A2, SIGMA, ri) == o.nnz
    assert np.array_equal(ref.Lp, o.Lp) and np.array_equal(ref.Li[:o.nnz], o.Li[:o.nnz])
    assert np.array_equal(ref.Lx[:o.nnz], o.Lx[:o.nnz]) and np.array_equal(ref.Dinv, o.Dinv)
    assert ldl_residual(o, kkt(P2, A2, ri)) < 1e-13


def test_coupling_other_than_minus_identity_is_refused():
    wl, P, A, ri = problem(3)
    N, nx, nu, ny, nt = wl.dims
    A2 = A.tolil()
    assert A2[ny, nu] == -1.0                                   # Aij(ny + 0, 0) of row block 0 against x_1
    A2[ny, nu] = -2.0
    A2 = A2.tocsc()
    assert ob.OracleRLDL(wl.dims).factor(P, A2, SIGMA, ri) == -3


@pytest.mark.parametrize("N,Nx", [(2, 1), (7, 3), (7, 6), (12, 5)])
def test_border_algebra_of_the_combined_variant_is_the_restart_block_of_the_recursion(N, Nx):
    """The reference's "combined" X / Z / Y variant (osqp_setup_combine_recursive :2359-2756, osqp_update_Z_horizon :2761-2856) keeps a
    finished factor and borders it: V^ = V L^-T D^-1 (compute_Vhat :253-292), Y^ = Y - V^ D V^' (:294-303), then factorises Y^.
    With the stage-interleaved order the bordered block is the cost block of stage Nx (block 2 Nx of Q0, C0, Q1, ...): its border
    rows V couple it to the factorised blocks 0 .. 2 Nx - 1 (only to the constraint block of stage Nx - 1).  The restatement of that
    algebra (orc_rldl_border) must give (1) V^ = the coupling rows L(2 Nx, :) of the full stage recursion and (2) Y^ = L_bb D_b L_bb',
    the Schur block the recursion forms and factorises when it reaches block 2 Nx -- which is why restarting the recursion at a block
    IS the Z / V^ / Y^ rebuild of a horizon change (what rldl_horizon.c's single store does)."""
    wl, P, A, ri = problem(N)
    o = ob.OracleRLDL(wl.dims)
    assert o.factor(P, A, SIGMA, ri) > 0
    Kp = kkt(P, A, ri)[np.ix_(o.perm, o.perm)]
    nx, nu, ny = wl.nx, wl.nu, wl.ny
    c0 = nu + (nx + ny) + (Nx - 1) * (2 * nx + nu + ny)                 # first permuted index of block 2 Nx (cost block of stage Nx)
    s = nx if Nx == N else nx + nu
    L = o.L().tocsc()
    Lf = L[:c0, :c0].tocsc()                                            # the finished factor: blocks before the bordered one
    Lf.sort_indices()
    V, Y = Kp[c0:c0 + s, :c0], Kp[c0:c0 + s, c0:c0 + s]
    Vh, Yh = ob.rldl_border(Lf.indptr, Lf.indices, Lf.data, o.Dinv[:c0], V, Y)
    Lrow = L[c0:c0 + s, :c0].toarray()
    assert np.max(np.abs(Vh - Lrow)) <= 1e-12 * max(1.0, np.max(np.abs(Lrow)))
    prev = c0 - (nx + ny)
    assert np.count_nonzero(Vh[:, :prev]) == 0 and np.count_nonzero(Vh[:, prev:]) > 0   # the border reaches the previous constraint block only
    Lbb = L[c0:c0 + s, c0:c0 + s].toarray() + np.eye(s)
    Sb = Lbb @ np.diag(1.0 / o.Dinv[c0:c0 + s]) @ Lbb.T
    assert np.max(np.abs(Yh - Sb)) <= 1e-12 * np.max(np.abs(Sb))
    # and the algebra is the Schur complement it claims to be: Y^ = Y - V D_f^-1... against numpy
    Ld = Lf.toarray() + np.eye(c0)
    W = np.linalg.solve(Ld, V.T)
    assert np.max(np.abs(Yh - (Y - W.T @ np.diag(o.Dinv[:c0]) @ W))) <= 1e-11 * np.max(np.abs(Sb))

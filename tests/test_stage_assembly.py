"""CPU test of the stage-block assembly (C-ABI `rldl_setup_AP_matrices`, the layout of setup_AP_matrices,
src/recursive_ldl.c:1873-1970) against an independent numpy/scipy block assembly, and of the source maps that a
batched update_AP_matrices (:1675-1778) relies on.  The recursive path has no reference fixture ("parity unpinned",
DESIGN.md section 5): what is checked is the algebra stated in SURVEY.md Appendix B."""
import ctypes as C

import numpy as np
import pytest
from scipy import sparse

import osqp_recursive_ldl_amd as R
from osqp_recursive_ldl_amd import _lib
from osqp_recursive_ldl_amd.linsys import CscPattern


def _csc_to_scipy(M):
    n, m, nz = M.n, M.m, M.p[M.n]
    p = np.array([M.p[i] for i in range(n + 1)]); i = np.array([M.i[k] for k in range(nz)])
    x = np.array([M.x[k] for k in range(nz)])
    return sparse.csc_matrix((x, i, p), shape=(m, n))


@pytest.mark.parametrize("N", [1, 2, 5])
def test_setup_AP_matrices_matches_block_assembly(N):
    wl = R.workloads.MPCStageQPs(N=max(N, 2)) if N >= 2 else None
    if wl is None:
        wl = R.workloads.MPCStageQPs(N=2)          # blocks only; assemble N = 1 by hand below
    dims = _lib.StageDims(N, wl.nx, wl.nu, wl.ny, wl.nt)
    blocks = [sparse.triu(sparse.csc_matrix(wl.Q0), format="csc"), sparse.triu(sparse.csc_matrix(wl.Qi), format="csc"),
              sparse.triu(sparse.csc_matrix(wl.QN), format="csc"), sparse.csc_matrix(wl.A0), sparse.csc_matrix(wl.Ai),
              sparse.csc_matrix(wl.Aij), sparse.csc_matrix(wl.AN)]
    holders = [CscPattern(b) for b in blocks]
    capP = (N - 1) * blocks[1].nnz + blocks[0].nnz + blocks[2].nnz
    capA = N * (blocks[4].nnz + blocks[5].nnz) + blocks[3].nnz + blocks[6].nnz
    src = [np.full(max(c, 1), -1, np.int64) for c in (capP, capP, capP, capA, capA, capA)]
    Pp, Ap = C.POINTER(_lib.Csc)(), C.POINTER(_lib.Csc)()
    rc = R.lib().rldl_setup_AP_matrices(C.byref(dims), *[h.ref for h in holders], C.byref(Pp), C.byref(Ap),
                                        *[a.ctypes.data_as(_lib.IP) for a in src])
    assert rc == 0
    P, A = _csc_to_scipy(Pp.contents), _csc_to_scipy(Ap.contents)
    # independent assembly
    nx, nu, ny, nt = wl.nx, wl.nu, wl.ny, wl.nt
    n, m = N * (nx + nu), N * (nx + ny) + nt
    Pd, Ad = np.zeros((n, n)), np.zeros((m, n))
    col0 = lambda k: 0 if k == 0 else nu + (k - 1) * (nx + nu)
    Pd[:nu, :nu] = wl.Q0
    Ad[:ny + nx, :nu] = wl.A0
    for k in range(1, N):
        c, r = col0(k), k * (nx + ny)
        Pd[c:c + nx + nu, c:c + nx + nu] = wl.Qi
        Ad[r:r + ny + nx, c:c + nx + nu] = wl.Ai
        Ad[r - (nx + ny):r, c:c + nx + nu] = wl.Aij
    c, r = col0(N), N * (nx + ny)
    Pd[c:c + nx, c:c + nx] = wl.QN
    Ad[r:r + nt, c:c + nx] = wl.AN
    Ad[r - (nx + ny):r, c:c + nx] = wl.Aij[:, :nx]
    assert P.shape == (n, n) and A.shape == (m, n)
    assert np.array_equal(P.toarray(), np.triu(Pd))
    assert np.array_equal(A.toarray(), Ad)
    # source maps: every stored value can be re-created from its block
    Pk, Ps, Pe, Ak, As, Ae = src
    Pblocks = {0: blocks[0], 1: blocks[1], 2: blocks[2]}
    Ablocks = {0: blocks[3], 1: blocks[4], 2: blocks[5], 3: blocks[6]}
    Pc, Ac = sparse.csc_matrix(P), sparse.csc_matrix(A)
    Px = np.array([Pp.contents.x[k] for k in range(Pp.contents.p[n])])
    Ax = np.array([Ap.contents.x[k] for k in range(Ap.contents.p[n])])
    assert all(Px[k] == Pblocks[Pk[k]].data[Pe[k]] for k in range(len(Px)))
    assert all(Ax[k] == Ablocks[Ak[k]].data[Ae[k]] for k in range(len(Ax)))
    assert (Ps[:len(Px)] >= 0).all() and (Ps[:len(Px)] <= N).all() and (As[:len(Ax)] <= N).all()
    if N >= 2:
        ref = R.workloads.MPCStageQPs(N=N)
        assert np.array_equal(P.toarray() != 0, ref.P_pattern.toarray() != 0)
        assert np.array_equal(A.toarray() != 0, ref.A_pattern.toarray() != 0)
    R.lib().rldl_csc_free(Pp); R.lib().rldl_csc_free(Ap)
    del Pc, Ac


def test_setup_AP_matrices_rejects_inconsistent_blocks():
    wl = R.workloads.MPCStageQPs(N=2)
    dims = _lib.StageDims(3, wl.nx, wl.nu, wl.ny, wl.nt + 1)      # nt does not match AN
    blocks = [CscPattern(sparse.triu(sparse.csc_matrix(b), format="csc") if sq else sparse.csc_matrix(b))
              for b, sq in ((wl.Q0, 1), (wl.Qi, 1), (wl.QN, 1), (wl.A0, 0), (wl.Ai, 0), (wl.Aij, 0), (wl.AN, 0))]
    Pp, Ap = C.POINTER(_lib.Csc)(), C.POINTER(_lib.Csc)()
    rc = R.lib().rldl_setup_AP_matrices(C.byref(dims), *[h.ref for h in blocks], C.byref(Pp), C.byref(Ap),
                                        None, None, None, None, None, None)
    assert rc == 1 and not Pp and not Ap

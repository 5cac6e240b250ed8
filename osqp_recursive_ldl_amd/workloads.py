"""Synthetic workloads of SURVEY.md section 8(d) (numpy only; no GPU, no oracle).

config 2 / metric shape : random sparse QPs, n=50, m=100, density 0.15, shared sparsity pattern
config 3 / config 5     : MPC stage blocks (quadcopter model of the reference's docs/examples/mpc.rst:31-72,
                          data only), N=20, nx=12, nu=4, ny=10, nt=12, assembled the way
                          setup_AP_matrices does (src/recursive_ldl.c:1873-1970).
"""
import numpy as np
from scipy import sparse


# ------------------------------------------------------------------ random sparse QPs ----
class SharedPatternQPs:
    """B instances of  min 1/2 x'Px + q'x  s.t. l <= Ax <= u  sharing one sparsity pattern.

    P = triu(M M' + I) with M = sprandn(n, n, density); A = sprandn(m, n, density); q ~ N(0,1);
    l = -3 + N(0,1), u = 3 + N(0,1) (the style of tests/primal_infeasibility/generate_problem.py:10-20).
    The pattern comes from `pattern_seed`; the VALUES of instance b from PCG64(1000 + seed0 + b).
    """

    def __init__(self, n=50, m=100, density=0.15, pattern_seed=1000):
        self.n, self.m = n, m
        prg = np.random.Generator(np.random.PCG64(pattern_seed))
        M = sparse.random(n, n, density=density, format="csc", random_state=prg)
        A = sparse.random(m, n, density=density, format="csc", random_state=prg)
        M.sort_indices(); A.sort_indices()
        self.Mp, self.Mi = M.indptr.copy(), M.indices.copy()
        self.A_pattern = sparse.csc_matrix((np.ones(A.nnz), A.indices, A.indptr), shape=(m, n))
        # structural pattern of triu(M M' + I) and the product map: P[slot] = sum_k M[a_k] * M[b_k] (+1 on the diagonal)
        pairs = {}
        for k in range(n):
            rows = self.Mi[self.Mp[k]:self.Mp[k + 1]]
            for ia, ra in enumerate(rows):
                for ib, rb in enumerate(rows):
                    if ra <= rb:
                        pairs.setdefault((rb, ra), []).append((self.Mp[k] + ia, self.Mp[k] + ib))
        for i in range(n):
            pairs.setdefault((i, i), [])
        keys = sorted(pairs.keys())                     # sorted by column then row = CSC order
        self.P_cols = np.array([c for c, r in keys]); self.P_rows = np.array([r for c, r in keys])
        Pp = np.zeros(n + 1, np.int64)
        np.add.at(Pp, self.P_cols + 1, 1)
        self.Pp = np.cumsum(Pp)
        self.P_pattern = sparse.csc_matrix((np.ones(len(keys)), self.P_rows, self.Pp), shape=(n, n))
        self._pa = np.array([a for k in keys for a, b in pairs[k]], np.int64)
        self._pb = np.array([b for k in keys for a, b in pairs[k]], np.int64)
        cnt = np.array([len(pairs[k]) for k in keys], np.int64)
        self._starts = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        self._empty = cnt == 0
        self._diag = self.P_cols == self.P_rows
        self.nnzP, self.nnzA, self.nnzM = len(keys), A.nnz, M.nnz

    def values(self, B, seed0=0):
        """(Px[B,nnzP], Ax[B,nnzA], q[B,n], l[B,m], u[B,m]) float64 host arrays."""
        n, m = self.n, self.m
        Px = np.empty((B, self.nnzP)); Ax = np.empty((B, self.nnzA))
        q = np.empty((B, n)); l = np.empty((B, m)); u = np.empty((B, m))
        for b in range(B):
            rg = np.random.Generator(np.random.PCG64(1000 + seed0 + b))
            Mv = rg.standard_normal(self.nnzM)
            prod = Mv[self._pa] * Mv[self._pb]
            if prod.size:
                px = np.add.reduceat(prod, np.minimum(self._starts, prod.size - 1))
                px[self._empty] = 0.0
            else:
                px = np.zeros(self.nnzP)
            px[self._diag] += 1.0
            Px[b] = px
            Ax[b] = rg.standard_normal(self.nnzA)
            q[b] = rg.standard_normal(n)
            l[b] = -3 + rg.standard_normal(m)
            u[b] = 3 + rg.standard_normal(m)
        l = np.minimum(l, u - 0.1)
        return Px, Ax, q, l, u

    def instance(self, b, seed0=0):
        """scipy matrices of instance b (for the CPU oracle / single-instance API)."""
        Px, Ax, q, l, u = self.values(1, seed0 + b)
        P = sparse.csc_matrix((Px[0], self.P_pattern.indices, self.P_pattern.indptr), shape=(self.n, self.n))
        A = sparse.csc_matrix((Ax[0], self.A_pattern.indices, self.A_pattern.indptr), shape=(self.m, self.n))
        return P, q[0], A, l[0], u[0]


# ------------------------------------------------------------------ MPC stage blocks ----
_AD = np.array([
    [1., 0., 0., 0., 0., 0., 0.1, 0., 0., 0., 0., 0.],
    [0., 1., 0., 0., 0., 0., 0., 0.1, 0., 0., 0., 0.],
    [0., 0., 1., 0., 0., 0., 0., 0., 0.1, 0., 0., 0.],
    [0.0488, 0., 0., 1., 0., 0., 0.0016, 0., 0., 0.0992, 0., 0.],
    [0., -0.0488, 0., 0., 1., 0., 0., -0.0016, 0., 0., 0.0992, 0.],
    [0., 0., 0., 0., 0., 1., 0., 0., 0., 0., 0., 0.0992],
    [0., 0., 0., 0., 0., 0., 1., 0., 0., 0., 0., 0.],
    [0., 0., 0., 0., 0., 0., 0., 1., 0., 0., 0., 0.],
    [0., 0., 0., 0., 0., 0., 0., 0., 1., 0., 0., 0.],
    [0.9734, 0., 0., 0., 0., 0., 0.0488, 0., 0., 0.9846, 0., 0.],
    [0., -0.9734, 0., 0., 0., 0., 0., -0.0488, 0., 0., 0.9846, 0.],
    [0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0.9846]])
_BD = np.array([
    [0., -0.0726, 0., 0.0726], [-0.0726, 0., 0.0726, 0.], [-0.0152, 0.0152, -0.0152, 0.0152],
    [0., -0.0006, 0., 0.0006], [0.0006, 0., -0.0006, 0.], [0.0106, 0.0106, 0.0106, 0.0106],
    [0., -1.4512, 0., 1.4512], [-1.4512, 0., 1.4512, 0.], [-0.3049, 0.3049, -0.3049, 0.3049],
    [0., -0.0236, 0., 0.0236], [0.0236, 0., -0.0236, 0.], [0.2107, 0.2107, 0.2107, 0.2107]])
_QDIAG = np.array([0.01, 0.01, 10., 10., 10., 10., 0.01, 0.01, 0.01, 5., 5., 5.])


class MPCStageQPs:
    """Stage-structured QPs in the layout the reference's recursive path expects
    (src/recursive_ldl.c:1898-1961): variables [u0 | x1,u1 | ... | x_{N-1},u_{N-1} | x_N], row block k =
    [ny inequality rows ; nx dynamics rows], terminal block of nt rows; A is block upper bidiagonal with
    A_k on the diagonal and Aij = [0 0; -I 0] on the super-diagonal."""

    def __init__(self, N=20, nx=12, nu=4, ny=10, nt=12):
        assert nx == 12 and nu == 4 and ny == 10 and nt == 12, "quadcopter data is fixed-size"
        self.N, self.nx, self.nu, self.ny, self.nt = N, nx, nu, ny, nt
        self.n = N * (nx + nu)
        self.m = N * (nx + ny) + nt
        self.dims = (N, nx, nu, ny, nt)
        Q, R = np.diag(_QDIAG), 0.1 * np.eye(nu)
        self.Q0, self.Qi, self.QN = R, np.block([[Q, np.zeros((nx, nu))], [np.zeros((nu, nx)), R]]), Q
        # inequality rows: 6 bounded states (angles, height, velocities) + 4 inputs
        Cx = np.zeros((6, nx)); Cx[[0, 1, 2, 3, 4, 5], [0, 1, 5, 6, 7, 8]] = 1.0
        G0 = np.array([[1., -1, 0, 0], [0, 1, -1, 0], [0, 0, 1, -1], [1, 0, 0, -1], [1, 1, 1, 1], [1, -1, 1, -1]])
        self.A0 = np.vstack([G0, np.eye(nu), _BD])                                         # (ny+nx) x nu
        self.Ai = np.vstack([np.hstack([Cx, np.zeros((6, nu))]), np.hstack([np.zeros((nu, nx)), np.eye(nu)]),
                             np.hstack([_AD, _BD])])                                        # (ny+nx) x (nx+nu)
        self.Aij = np.zeros((ny + nx, nx + nu)); self.Aij[ny:, :nx] = -np.eye(nx)
        self.AN = np.eye(nx)                                                                # nt x nx
        self._build_patterns()

    def _col0(self, k):
        return 0 if k == 0 else self.nu + (k - 1) * (self.nx + self.nu)

    def _assemble(self, Q0, Qi_list, QN, A0, Ai_list, AN):
        N, nx, nu, ny, nt = self.dims
        P = sparse.lil_matrix((self.n, self.n)); A = sparse.lil_matrix((self.m, self.n))
        P[:nu, :nu] = Q0
        A[:ny + nx, :nu] = A0
        for k in range(1, N):
            c = self._col0(k); r = k * (nx + ny)
            P[c:c + nx + nu, c:c + nx + nu] = Qi_list[k - 1]
            A[r:r + ny + nx, c:c + nx + nu] = Ai_list[k - 1]
            A[r - (nx + ny):r, c:c + nx + nu] = self.Aij          # coupling of the previous row block to x_k
        c = self._col0(N); r = N * (nx + ny)
        P[c:c + nx, c:c + nx] = QN
        A[r:r + nt, c:c + nx] = AN
        A[r - (nx + ny):r, c:c + nx] = self.Aij[:, :nx]
        P = sparse.triu(P.tocsc(), format="csc"); A = A.tocsc()
        P.sort_indices(); A.sort_indices()
        return P, A

    def _build_patterns(self):
        N = self.N
        P, A = self._assemble(self.Q0, [self.Qi] * (N - 1), self.QN, self.A0, [self.Ai] * (N - 1), self.AN)
        self.P_pattern, self.A_pattern = P, A
        self.nnzP, self.nnzA = P.nnz, A.nnz
        self._P0x, self._A0x = P.data.copy(), A.data.copy()
        # which stage each stored value belongs to (for config 5: perturb stages >= k only)
        cols_stage = np.zeros(self.n, np.int64)
        for k in range(1, N + 1):
            cols_stage[self._col0(k):] = k
        self.P_stage = cols_stage[np.repeat(np.arange(self.n), np.diff(P.indptr))]
        self.A_stage = cols_stage[np.repeat(np.arange(self.n), np.diff(A.indptr))]

    def values(self, B, seed0=0, rel=0.01):
        """Per-instance perturbation of the nonzeros by 1 + rel*N(0,1) (the -I couplings stay exact)."""
        Px = np.empty((B, self.nnzP)); Ax = np.empty((B, self.nnzA))
        q = np.empty((B, self.n)); l = np.empty((B, self.m)); u = np.empty((B, self.m))
        exact = self._A0x == -1.0
        N, nx, nu, ny, nt = self.dims
        for b in range(B):
            rg = np.random.Generator(np.random.PCG64(5000 + seed0 + b))
            Px[b] = self._P0x * (1 + rel * rg.standard_normal(self.nnzP))
            ax = self._A0x * (1 + rel * rg.standard_normal(self.nnzA))
            ax[exact] = -1.0
            Ax[b] = ax
            q[b] = 0.1 * rg.standard_normal(self.n)
            lo = np.empty(self.m); up = np.empty(self.m)
            x0 = 0.1 * rg.standard_normal(nx)
            for k in range(N):
                r = k * (nx + ny)
                lo[r:r + ny] = -1.0 - rg.random(ny); up[r:r + ny] = 1.0 + rg.random(ny)
                lo[r + ny:r + ny + nx] = up[r + ny:r + ny + nx] = (-_AD @ x0 if k == 0 else 0.0)   # dynamics: equality rows
            r = N * (nx + ny)
            lo[r:] = -2.0; up[r:] = 2.0
            l[b], u[b] = lo, up
        return Px, Ax, q, l, u

    def instance(self, b, seed0=0):
        Px, Ax, q, l, u = self.values(1, seed0 + b)
        P = sparse.csc_matrix((Px[0], self.P_pattern.indices, self.P_pattern.indptr), shape=(self.n, self.n))
        A = sparse.csc_matrix((Ax[0], self.A_pattern.indices, self.A_pattern.indptr), shape=(self.m, self.n))
        return P, q[0], A, l[0], u[0]


def stage_permutation(N, nx, nu, ny, nt):
    """Closed-form interleave Q0,C0,Q1,C1,...,QN,CN (src/recursive_ldl.c:1350-1362), numpy restatement
    used by tests to pin the C implementation."""
    nvar = N * (nx + nu)
    perm, pc, ac = [], 0, nvar
    perm += list(range(pc, pc + nu)); pc += nu
    perm += list(range(ac, ac + nx + ny)); ac += nx + ny
    for _ in range(N - 1):
        perm += list(range(pc, pc + nx + nu)); pc += nx + nu
        perm += list(range(ac, ac + nx + ny)); ac += nx + ny
    perm += list(range(pc, pc + nx)); pc += nx
    perm += list(range(ac, ac + nt)); ac += nt
    return np.array(perm, np.int64)

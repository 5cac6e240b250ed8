"""Shared test helpers: golden-fixture loader and the synthetic workload generators of
SURVEY.md section 8(d) (config 2: random sparse QPs; config 3: MPC stage blocks)."""
import json
import os

import numpy as np
from scipy import sparse

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _dec(v):
    if isinstance(v, dict) and v.get("__csc__"):
        return sparse.csc_matrix((np.array(v["x"], float), np.array(v["i"], np.int64), np.array(v["p"], np.int64)),
                                 shape=(v["m"], v["n"]))
    if isinstance(v, dict) and v.get("__vec__"):
        return np.array([float(t) for t in v["x"]], dtype=float)
    if isinstance(v, dict):
        return {k: _dec(t) for k, t in v.items()}
    if v in ("inf", "-inf"):
        return float(v)
    return v


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return _dec(json.load(f))


def osqp_inf(v):
    """The reference clips infinities to +-OSQP_INFTY (codegen_utils writes +-1e30-ish; constants.h:99)."""
    return np.clip(np.asarray(v, float), -1e30, 1e30)


def full_kkt(P_triu, A, sigma, rho_vec):
    """Dense symmetric KKT [P+sigma I, A'; A, -diag(1/rho)]."""
    P = sparse.csc_matrix(P_triu)
    Pf = (P + sparse.triu(P, 1).T).toarray()
    A = sparse.csc_matrix(A).toarray()
    n, m = Pf.shape[0], A.shape[0]
    K = np.zeros((n + m, n + m))
    K[:n, :n] = Pf + sigma * np.eye(n)
    K[:n, n:] = A.T
    K[n:, :n] = A
    K[n:, n:] = -np.diag(1.0 / np.asarray(rho_vec, float)) if m else 0
    return K


def dense_from_L(Lp, Li, Lx, N):
    L = np.eye(N)
    for j in range(N):
        for p in range(Lp[j], Lp[j + 1]):
            L[Li[p], j] = Lx[p]
    return L


def dense_symbolic(Kperm_upper_pattern):
    """Independent symbolic Cholesky of a permuted pattern (dense boolean elimination):
    returns (etree, Lnz, list of row sets per column)."""
    N = Kperm_upper_pattern.shape[0]
    S = (Kperm_upper_pattern + Kperm_upper_pattern.T) != 0
    S = S.copy()
    cols = []
    etree = -np.ones(N, np.int64)
    for j in range(N):
        rows = [r for r in range(j + 1, N) if S[r, j]]
        cols.append(rows)
        if rows:
            etree[j] = rows[0]
        for a in rows:
            for b in rows:
                S[a, b] = True
    Lnz = np.array([len(c) for c in cols], np.int64)
    return etree, Lnz, cols


def random_qp(seed, n=50, m=100, density=0.15, pattern_seed=None):
    """Config-2 instance: P = triu(M M' + I), A = sprandn, q ~ N(0,1),
    l = -3 + N(0,1) clipped below u, u = 3 + N(0,1).  With pattern_seed set, the sparsity
    pattern comes from that seed and only the VALUES depend on `seed` (shared-pattern batch)."""
    prg = np.random.Generator(np.random.PCG64(pattern_seed if pattern_seed is not None else seed))
    M = sparse.random(n, n, density=density, format="csc", random_state=prg)
    A = sparse.random(m, n, density=density, format="csc", random_state=prg)
    M.sort_indices(); A.sort_indices()
    rg = np.random.Generator(np.random.PCG64(1000 + seed))
    M = sparse.csc_matrix((rg.standard_normal(M.nnz), M.indices, M.indptr), shape=M.shape)
    A = sparse.csc_matrix((rg.standard_normal(A.nnz), A.indices, A.indptr), shape=A.shape)
    Pfull = (M @ M.T + sparse.eye(n)).tocsc()
    # keep the pattern value-independent: take the structural pattern of M M' + I
    P = sparse.triu(Pfull, format="csc")
    P.sort_indices()
    q = rg.standard_normal(n)
    l = -3 + rg.standard_normal(m)
    u = 3 + rg.standard_normal(m)
    l = np.minimum(l, u - 0.1)
    return P, q, A, l, u


def shared_pattern_batch(B, n=50, m=100, density=0.15, pattern_seed=1000):
    """B config-2 instances sharing one sparsity pattern; returns (P0, A0, Px[B,nnzP], Ax[B,nnzA], q, l, u)."""
    P0, q0, A0, l0, u0 = random_qp(0, n, m, density, pattern_seed)
    nnzP, nnzA = P0.nnz, A0.nnz
    Px = np.zeros((B, nnzP)); Ax = np.zeros((B, nnzA))
    q = np.zeros((B, n)); l = np.zeros((B, m)); u = np.zeros((B, m))
    for b in range(B):
        P, qq, A, ll, uu = random_qp(b, n, m, density, pattern_seed)
        assert P.nnz == nnzP and A.nnz == nnzA and (P.indices == P0.indices).all()
        Px[b], Ax[b], q[b], l[b], u[b] = P.data, A.data, qq, ll, uu
    return P0, A0, Px, Ax, q, l, u


# ---- product form of the block tri-solve (csrc/rldl_recursive.c: build_prod_tiles; kernels: stage_prod_solve) ----
def prod_block_starts(dims):
    """First permuted index of every stage block (compute_permutations order, src/recursive_ldl.c:1350-1362)."""
    N, nx, nu, ny, nt = dims
    bs = [0, nu]
    for _ in range(1, N):
        bs.append(bs[-1] + ny + nx); bs.append(bs[-1] + nx + nu)
    bs.append(bs[-1] + ny + nx); bs.append(bs[-1] + nx); bs.append(bs[-1] + nt)
    return np.array(bs)


def prod_emulate(pr, Dinv, b):
    """The two passes exactly as stage_prod_solve runs them (per step of the sequence, per lane), on the permuted right-hand side b."""
    xs = b.copy()
    tab, Ti, prog = pr["tab"], pr["Ti"], pr["prog"]
    state = dict(acc=np.zeros(64), own=np.zeros(64))
    NS = pr["steps"]

    def group(step, fwd):
        d = [int(v) & 0xffffffff for v in prog[step]]
        ti0, wo, base, fl = d[:4]
        masks = [d[4 + 2 * j] | (d[5 + 2 * j] << 32) for j in range(4)]
        w = tab[wo:wo + 64].astype(np.int64)
        row = (base >> 16) // 8 + ((w >> 24) & 31)
        has = (w >> 29) & 1
        if fwd and (fl & 1):
            state["acc"][:] = 0.0
        if not fwd and (fl & 2):
            state["own"] = xs[row].copy()
        off = ti0
        for j in range(4):
            col = (base & 0xffff) // 8 + ((w >> (5 * j)) & 31)
            for lane in range(64):
                if (masks[j] >> lane) & 1:
                    assert (w[lane] >> (20 + j)) & 1
                    v = Ti[off]; off += 1                          # entries of a step in lane order
                    if fwd:
                        state["acc"][lane] += v * xs[col[lane]]
                    else:
                        xs[col[lane]] -= v * state["own"][lane]
        if fwd and (fl & 2):
            for lane in range(64):
                if has[lane]:
                    xs[row[lane]] -= state["acc"][lane]

    for st in range(NS // 2):
        group(st, True)
    xs *= Dinv
    for st in range(NS // 2, NS):
        group(st, False)
    return xs

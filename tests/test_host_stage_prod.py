"""Tables of the product form of the block tri-solve (csrc/rldl_recursive.c: build_prod_tiles -> stage_prod_solve, k_stage_invert),
checked on the CPU through the host-only export: the closure of every diagonal block's pattern covers its inverse, every entry
of L has exactly one place in the tiles, and a numpy emulation of the two passes -- per step of the kernel's sequence, per lane,
masks and table words as the kernel reads them -- solves L D L' x = b for random values on the pattern.  What QDLDL_solve does
(qdldl_interface.c:538-585) on the factor of LDL_factorize_recursive (src/recursive_ldl.c:1139-1318)."""
import numpy as np
import pytest
from scipy import sparse
from scipy.linalg import solve_triangular

from helpers import prod_block_starts, prod_emulate


@pytest.mark.parametrize("N", [1, 2, 5, 20, 45])
def test_tables_cover_the_factor_and_the_emulated_passes_solve(N):
    import osqp_recursive_ldl_amd as R
    wl = R.workloads.MPCStageQPs(N=N)
    pr = R.linsys.stage_prod_export(wl.dims, wl.P_pattern, wl.A_pattern)
    assert pr is not None
    sym = pr["sym"]
    Nk, nnzL = wl.n + wl.m, sym["nnzL"]
    rng = np.random.default_rng(100 + N)
    Lx = 0.3 * rng.standard_normal(nnzL)
    Dinv = 1.0 / (rng.uniform(0.5, 2.0, Nk) * np.where(rng.random(Nk) < 0.5, -1.0, 1.0))       # quasi-definite: both signs
    L = sparse.csc_matrix((Lx, sym["Li"], sym["Lp"]), shape=(Nk, Nk)).toarray() + np.eye(Nk)
    bs = prod_block_starts(wl.dims)
    assert len(bs) - 1 == pr["nb"]
    ld = pr["ld"]
    slot_to_p = np.full(int(pr["LtoS"].max()) + 1, -1, np.int64)
    slot_to_p[pr["LtoS"]] = np.arange(nnzL)
    col_of_p = np.repeat(np.arange(Nk), np.diff(sym["Lp"]))
    Ti = np.zeros(len(pr["src"]))
    seen = np.zeros(nnzL, bool)
    for b in range(pr["nb"]):
        td, tc = pr["blk"][b]
        s0, s1 = bs[b], bs[b + 1]
        if td >= 0:
            ti0, E, kind, g0 = [int(v) for v in pr["tinfo"][td]]
            assert kind == 0
            Xi = np.linalg.inv(L[s0:s1, s0:s1])
            src = pr["src"][ti0:ti0 + E].astype(np.int64)
            r, c = src // ld, src % ld
            assert np.all(r > c) and np.all(r < s1 - s0) and len(set(zip(r.tolist(), c.tolist()))) == E
            Ti[ti0:ti0 + E] = -Xi[r, c]
            rest = np.tril(Xi, -1); rest[r, c] = 0.0
            assert np.max(np.abs(rest), initial=0.0) <= 1e-12, "the pattern of a diagonal tile misses an entry of the inverse"
        else:
            assert np.count_nonzero(np.tril(L[s0:s1, s0:s1], -1)) == 0
        if tc >= 0:
            ti0, E, kind, g0 = [int(v) for v in pr["tinfo"][tc]]
            assert kind == 1
            p = slot_to_p[pr["src"][ti0:ti0 + E].astype(np.int64)]
            assert np.all(p >= 0) and not np.any(seen[p])
            seen[p] = True
            rows, cols = sym["Li"][p], col_of_p[p]
            assert np.all((cols >= s0) & (cols < s1) & (rows >= s1) & (rows < bs[b + 2]))            # entries of L(b + 1, b)
            Ti[ti0:ti0 + E] = Lx[p]
    in_coupling = np.array([np.searchsorted(bs, sym["Li"][p], side="right") != np.searchsorted(bs, col_of_p[p], side="right") for p in range(nnzL)])
    assert np.array_equal(seen, in_coupling)                                                        # every coupling entry exactly once
    pr = dict(pr, Ti=Ti)
    # lane masks and table words agree, steps of the sequence are a multiple of four groups: forward half then backward half
    assert pr["steps"] % 2 == 0 and len(pr["prog"]) == pr["steps"]
    for trial in range(2):
        bp = rng.standard_normal(Nk)
        ref = solve_triangular(L.T, Dinv * solve_triangular(L, bp, lower=True, unit_diagonal=True), lower=False, unit_diagonal=True)
        emu = prod_emulate(pr, Dinv, bp)
        assert np.max(np.abs(emu - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref)))


def test_patterns_that_are_not_stage_structured_are_refused():
    import osqp_recursive_ldl_amd as R
    wl = R.workloads.SharedPatternQPs(n=20, m=32, density=0.3, pattern_seed=1)
    assert R.linsys.stage_prod_export((1, 12, 8, 10, 10), wl.P_pattern, wl.A_pattern) is None   # dims that do not describe this pattern

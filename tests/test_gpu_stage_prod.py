"""Product form of the block tri-solve on stage patterns (k_stage_invert + stage_prod_solve, csrc/rldl_kernels.hip): what
QDLDL_solve does on the factor of LDL_factorize_recursive (src/recursive_ldl.c:1139-1318, qdldl_interface.c:538-585), with
every diagonal block of L replaced by its inverse at factor time.  Three links are checked separately, so that a failure
names its place: (1) the tile values k_stage_invert wrote against numpy inverses of the exported L blocks, (2) a numpy
emulation of the two passes driven by the exported tables against a triangular solve with the exported L, D, (3) the
kernel's solve against that emulation and against the dense KKT solve."""
import os

import numpy as np
import pytest
from scipy import sparse
from scipy.linalg import solve_triangular

from helpers import prod_block_starts as block_starts, prod_emulate as emulate

torch = pytest.importorskip("torch")
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(any(os.environ.get(k) for k in ("RLDL_NO_STAGE_PROD", "RLDL_NO_STAGE_FACTOR")),
                                 reason="the product tri-solve is switched off")]
SIGMA = 1e-6


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to("cuda:0")


@pytest.mark.parametrize("tables", ["K-tiles", "L-tiles"])
@pytest.mark.parametrize("N", [1, 3, 20, 45])
def test_tiles_tables_and_solve(N, tables, monkeypatch):
    """tables = "L-tiles": the round-2 form (RLDL_PROD_V1=1: coupling tiles hold L(b+1, b), one use of every tile per pass);
    "K-tiles": the default since round 3 (stage_prod_solve2: coupling tiles hold the KKT matrix's own coupling blocks K(b+1, b) and
    every diagonal tile is applied twice per pass: L(b+1, b) = K(b+1, b) L_bb^-T D_b^-1 is never streamed)."""
    import osqp_recursive_ldl_amd as R
    if tables == "L-tiles":
        monkeypatch.setenv("RLDL_PROD_V1", "1")
    elif os.environ.get("RLDL_PROD_V1"):
        pytest.skip("RLDL_PROD_V1 is set: the K-tile tables are off")
    wl = R.workloads.MPCStageQPs(N=N)
    B = 2
    Px, Ax, q, l, u = wl.values(B)
    rho = np.where(np.abs(u - l) < 1e-4, 100.0, 0.1) * (1.0 + 0.1 * np.arange(wl.m) / wl.m)
    ls = R.BatchLinsys.recursive(wl.dims, wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), SIGMA, dev(rho))
    assert ls.status == 0
    sym = ls.export_symbolic()
    Nk = wl.n + wl.m
    bs = block_starts(wl.dims)
    rng = np.random.default_rng(5)
    rhs = rng.standard_normal((B, Nk))
    out = ls.solve(dev(rhs.copy())).cpu().numpy()
    for inst in range(B):
        pr = ls.export_prod(inst)
        assert pr is not None, "stage handle without product tables"
        assert pr["mode"] == (1 if tables == "L-tiles" else 2)
        f = ls.export_factor(inst)
        L = sparse.csc_matrix((f["Lx"], sym["Li"], sym["Lp"]), shape=(Nk, Nk)).toarray() + np.eye(Nk)
        # (1) tile values: D tiles hold -(strictly lower part of L_bb^-1), C tiles hold L(b+1, b)
        ld = pr["ld"]
        for b in range(pr["nb"]):
            td, tc = pr["blk"][b]
            s0, s1 = bs[b], bs[b + 1]
            if td >= 0:
                ti0, E, kind, g0 = [int(v) for v in pr["tinfo"][td]]
                assert kind == 0
                Xi = np.linalg.inv(L[s0:s1, s0:s1])
                src = pr["src"][ti0:ti0 + E].astype(np.int64)
                tv = pr["Ti"][ti0:ti0 + E]
                if pr["mode"] == 2:                                    # pair layout: padding slots (source 0xffff) hold 0.0
                    assert np.all(tv[src == 0xffff] == 0.0)
                    tv, src = tv[src != 0xffff], src[src != 0xffff]
                want = -Xi[src // ld, src % ld]
                assert np.max(np.abs(tv - want)) <= 1e-11 * max(1.0, np.max(np.abs(Xi)))
                dense = np.zeros_like(Xi); dense[src // ld, src % ld] = Xi[src // ld, src % ld]
                assert np.max(np.abs(np.tril(Xi, -1) - dense)) <= 1e-13 * max(1.0, np.max(np.abs(Xi))), "inverse pattern misses an entry"
            if tc >= 0 and pr["mode"] == 1:
                ti0, E, kind, g0 = [int(v) for v in pr["tinfo"][tc]]
                assert kind == 1 and E == np.count_nonzero(sym["Li"][sym["Lp"][s0]:sym["Lp"][s1]] >= s1)
            if tc >= 0 and pr["mode"] == 2:                            # the coupling tile IS the KKT matrix's coupling block: values = Kx[src]
                ti0, E, kind, g0 = [int(v) for v in pr["tinfo"][tc]]
                src = pr["src"][ti0:ti0 + E].astype(np.int64)
                tv = pr["Ti"][ti0:ti0 + E]
                assert kind == 1 and np.all(tv[src == 0xffff] == 0.0)
                assert np.array_equal(tv[src != 0xffff], f["KKTx"][src[src != 0xffff]]) and len(set(src[src != 0xffff].tolist())) == int((src != 0xffff).sum())
        # (2) the emulation of the two passes against a triangular solve with the exported factor
        perm = sym["perm"]
        bp = rhs[inst][perm]
        ref = solve_triangular(L.T, f["Dinv"] * solve_triangular(L, bp, lower=True, unit_diagonal=True), lower=False, unit_diagonal=True)
        scale = max(1.0, np.max(np.abs(ref)))
        if pr["mode"] == 1:
            emu = emulate(pr, f["Dinv"], bp)
            assert np.max(np.abs(emu - ref)) <= 1e-9 * scale
        # (3) the kernel: x_tilde / z_tilde epilogue undone (qdldl_interface.c:563-579)
        sol = np.empty(Nk); sol[perm] = ref
        got = out[inst].copy()
        got[wl.n:] = (got[wl.n:] - rhs[inst][wl.n:]) * rho[inst]
        assert np.max(np.abs(got - sol)) <= 1e-9 * scale
    ls.free()


@pytest.mark.parametrize("shape", ["mpc", "metric"])
def test_inverse_based_solves_keep_the_residual_small_under_extreme_rho(shape):
    """Both round-2 tri-solves multiply by explicitly inverted triangles (the tail of the arrowhead factor, the diagonal blocks of
    the stage factor) where QDLDL substitutes.  rho_vec spanning 1e-6 .. 1e6 (what adaptive rho and the equality-row rule can
    produce, osqp.c:1268-1319, auxil.c:88-91) stretches those triangles; the scaled residual of K x = b must stay at rounding level."""
    import osqp_recursive_ldl_amd as R
    from helpers import full_kkt
    rng = np.random.default_rng(17)
    B = 3
    if shape == "mpc":
        wl = R.workloads.MPCStageQPs(N=8)
        Px, Ax, q, l, u = wl.values(B)
        make = lambda rho: R.BatchLinsys.recursive(wl.dims, wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), SIGMA, dev(rho))
    else:
        wl = R.workloads.SharedPatternQPs()
        Px, Ax, q, l, u = wl.values(B)
        make = lambda rho: R.BatchLinsys(wl.P_pattern, wl.A_pattern, dev(Px), dev(Ax), SIGMA, dev(rho))
    for rho in (10.0 ** rng.uniform(-6, 6, (B, wl.m)), np.full((B, wl.m), 1e6), np.full((B, wl.m), 1e-6)):
        ls = make(rho)
        assert ls.status == 0
        rhs = rng.standard_normal((B, wl.n + wl.m))
        out = ls.solve(dev(rhs.copy())).cpu().numpy()
        for b in range(B):
            P, qq, A, ll, uu = wl.instance(b)
            K = full_kkt(sparse.triu(P), A, SIGMA, rho[b])
            x = out[b].copy()
            x[wl.n:] = (x[wl.n:] - rhs[b][wl.n:]) * rho[b]                    # nu from z_tilde (qdldl_interface.c:577-579)
            assert np.max(np.abs(K @ x - rhs[b])) <= 1e-10 * max(1.0, np.max(np.abs(x)))
        ls.free()

"""CPU tests (no GPU): the product's host-side symbolic analysis against the oracle and against
independent numpy/scipy restatements; C-ABI surface; loud failure without a device.

Integer quantities are compared EXACTLY (permutation given, etree, Lnz, Lp, Li, KKT pattern, maps).
"""
import ctypes as C
import re
import os

import numpy as np
import pytest
from scipy import sparse

import oracle_bindings as ob
from helpers import dense_symbolic, full_kkt, load_golden

import osqp_recursive_ldl_amd as R
from osqp_recursive_ldl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "osqp_rldl_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b([A-Za-z0-9_]*(?:hipldl|rldl_|osqp_batch|osqp_horizon|osqp_multi|osqp_dist|osqp_groups)[A-Za-z0-9_]*)\s*\(", hdr))
    names = {n for n in names if not n.startswith("c_")}
    assert len(names) >= 30
    L = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing
    assert set(_lib.EXPORTED) <= names | {"rldl_version"}
    assert b"gfx950" in R.lib().rldl_version()


def test_struct_prefix_matches_linsys_solver_layout():
    """include/types.h:298-319: type, solve, free, update_matrices, update_rho_vec, nthreads -- in this order."""
    f = [n for n, _ in _lib.HipldlSolver._fields_]
    assert f[:6] == ["type", "solve", "free", "update_matrices", "update_rho_vec", "nthreads"]
    assert _lib.HipldlSolver.solve.offset == 8 and _lib.HipldlSolver.nthreads.offset == 40


def _problem(seed, n=20, m=35, density=0.2):
    wl = R.workloads.SharedPatternQPs(n=n, m=m, density=density, pattern_seed=seed)
    return wl, wl.instance(0)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_symbolic_matches_oracle_given_the_same_permutation(seed):
    wl, (P, q, A, l, u) = _problem(seed)
    n, m = wl.n, wl.m
    s = R.symbolic_analyze(wl.P_pattern, wl.A_pattern)
    perm = s["perm"]
    assert sorted(perm.tolist()) == list(range(n + m))
    o = ob.OracleLinsys(P, A, 1e-6, np.full(m, 0.1), perm=perm)
    e = o.export()
    assert (e["P"] == perm).all()
    assert (e["etree"] == s["etree"]).all()
    assert (e["Lnz"] == s["Lnz"]).all()
    assert (e["Lp"] == s["Lp"]).all()
    assert (e["Li"] == s["Li"]).all()
    # KKT pattern: the oracle's csc_symperm output has the same entries per column (rows unsorted there)
    Kp, Ki, Kx = o.export_KKT()
    assert (Kp == s["KKTp"]).all()
    for j in range(n + m):
        assert sorted(Ki[Kp[j]:Kp[j + 1]].tolist()) == s["KKTi"][Kp[j]:Kp[j + 1]].tolist()
    # independent dense symbolic factorisation
    K = full_kkt(P, A, 1e-6, np.full(m, 0.1))
    Kperm = K[np.ix_(perm, perm)]
    et, Lnz, cols = dense_symbolic(np.triu(Kperm != 0))
    assert (et == s["etree"]).all() and (Lnz == s["Lnz"]).all()


@pytest.mark.parametrize("seed", [4, 5])
def test_scatter_maps_rebuild_the_permuted_kkt(seed):
    wl, (P, q, A, l, u) = _problem(seed, n=12, m=18, density=0.3)
    n, m = wl.n, wl.m
    sigma, rho = 0.37, np.linspace(0.5, 2.0, m)
    s = R.symbolic_analyze(wl.P_pattern, wl.A_pattern)
    Kx = np.zeros(s["nnzKKT"])
    Pd, Ad = sparse.csc_matrix(P), sparse.csc_matrix(A)
    Pd.sort_indices(); Ad.sort_indices()
    cols = np.repeat(np.arange(n), np.diff(Pd.indptr))
    Kx[:] = np.nan
    Kx[s["PtoKKT"]] = Pd.data + sigma * (Pd.indices == cols)
    Kx[s["AtoKKT"]] = Ad.data
    Kx[s["rhotoKKT"]] = -1.0 / rho
    Kx[np.isnan(Kx)] = sigma                       # sigma-only slots (P columns without a stored diagonal)
    Kdense = np.zeros((n + m, n + m))
    for j in range(n + m):
        for p in range(s["KKTp"][j], s["KKTp"][j + 1]):
            i = s["KKTi"][p]
            assert i <= j
            Kdense[i, j] = Kx[p]
    Kdense = Kdense + np.triu(Kdense, 1).T
    K = full_kkt(P, A, sigma, rho)
    perm = s["perm"]
    assert np.max(np.abs(Kdense - K[np.ix_(perm, perm)])) == 0.0


def test_form_kkt_pattern_against_reference_fixture():
    """tests/update_matrices fixture (reference generator): with the identity permutation the product's
    KKT pattern must be exactly triu of the fixture's KKT (test_update_matrices.h:37-48)."""
    d = load_golden("update_matrices")["data"]
    n, m = d["test_form_KKT_n"], d["test_form_KKT_m"]
    s = R.symbolic_analyze(d["test_form_KKT_Pu"], d["test_form_KKT_A"], perm=np.arange(n + m))
    ref = sparse.csc_matrix(d["test_form_KKT_KKTu"]); ref.sort_indices()
    assert (s["KKTp"] == ref.indptr).all() and (s["KKTi"] == ref.indices).all()
    Pu = sparse.csc_matrix(d["test_form_KKT_Pu"]); Pu.sort_indices()
    A = sparse.csc_matrix(d["test_form_KKT_A"]); A.sort_indices()
    cols = np.repeat(np.arange(n), np.diff(Pu.indptr))
    Kx = np.full(s["nnzKKT"], np.nan)
    Kx[s["PtoKKT"]] = Pu.data + d["test_form_KKT_sigma"] * (Pu.indices == cols)
    Kx[s["AtoKKT"]] = A.data
    Kx[s["rhotoKKT"]] = -1.0 / d["test_form_KKT_rho"]
    Kx[np.isnan(Kx)] = d["test_form_KKT_sigma"]
    assert np.max(np.abs(Kx - ref.data)) < 1e-14


def test_user_permutation_is_respected_and_validated():
    wl, (P, q, A, l, u) = _problem(6, n=8, m=10, density=0.3)
    N = wl.n + wl.m
    perm = np.random.default_rng(0).permutation(N)
    s = R.symbolic_analyze(wl.P_pattern, wl.A_pattern, perm=perm)
    assert (s["perm"] == perm).all()
    bad = perm.copy(); bad[0] = bad[1]
    with pytest.raises(ValueError):
        R.symbolic_analyze(wl.P_pattern, wl.A_pattern, perm=bad)


def test_rejects_lower_triangular_P():
    P = sparse.csc_matrix(np.array([[1.0, 0.0], [0.5, 1.0]]))
    A = sparse.csc_matrix(np.array([[1.0, 1.0]]))
    with pytest.raises(ValueError):
        R.symbolic_analyze(P, A)


def test_min_degree_reduces_fill_on_metric_shape():
    wl = R.workloads.SharedPatternQPs()          # n=50, m=100, density 0.15
    s = R.symbolic_analyze(wl.P_pattern, wl.A_pattern)
    nat = R.symbolic_analyze(wl.P_pattern, wl.A_pattern, perm=np.arange(150))
    assert s["nnzL"] < 0.5 * nat["nnzL"]
    assert 1500 < s["nnzL"] < 2600               # ~1970 with AMD on this shape
    assert s["etree_height"] <= 60


def test_stage_permutation_closed_form():
    for dims in [(20, 12, 4, 10, 12), (3, 2, 1, 2, 2), (1, 3, 2, 1, 3)]:
        N, nx, nu, ny, nt = dims
        tot = N * (nx + nu) + N * (nx + ny) + nt
        perm = np.zeros(tot, np.int64)
        R.lib().rldl_stage_permutation(N, nx, nu, ny, nt, perm.ctypes.data_as(_lib.IP))
        assert (perm == R.workloads.stage_permutation(*dims)).all()
        assert sorted(perm.tolist()) == list(range(tot))


def test_mpc_stage_order_gives_block_tridiagonal_band():
    wl = R.workloads.MPCStageQPs(N=5)
    perm = R.workloads.stage_permutation(*wl.dims)
    s = R.symbolic_analyze(wl.P_pattern, wl.A_pattern, perm=perm)
    # every L column reaches at most into the next two stage blocks
    band = max((s["Li"][s["Lp"][j]:s["Lp"][j + 1]].max() - j) for j in range(wl.n + wl.m) if s["Lnz"][j])
    assert band <= 2 * (wl.nx + wl.ny) + wl.nx + wl.nu
    assert s["etree_height"] > 0


def test_compute_entry_points_fail_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    wl, (P, q, A, l, u) = _problem(1, n=5, m=6)
    with pytest.raises(RuntimeError, match="no HIP device"):
        R.HipLDLSolver(P, A, 1e-6, np.full(wl.m, 0.1))
    h = C.c_void_p()
    Pc, Ac = R.CscPattern(wl.P_pattern), R.CscPattern(wl.A_pattern)
    rc = R.lib().rldl_batch_init(C.byref(h), 1, Pc.ref, Ac.ref, None, None, 1e-6, None, 0, None, None)
    assert rc == _lib.RLDL_NO_DEVICE_ERROR and not h.value


@pytest.mark.parametrize("case", ["arrowhead", "small", "tiny", "mpc", "unconstrained"])
def test_solve_plan_schedule_reproduces_the_reference_substitution(case):
    """The device schedule (groups, jagged-diagonal gathers, packed-triangle sweeps; csrc/rldl_plan.c) emulated on
    the CPU from the exported plan must equal the reference's column substitution (src/recursive_ldl.c:62-116) on
    the oracle's factor -- and every L entry must own exactly one storage slot."""
    from osqp_recursive_ldl_amd.linsys import plan_emulate_solve, plan_export
    perm = None
    if case == "arrowhead":
        wl = R.workloads.SharedPatternQPs()
    elif case == "small":
        wl = R.workloads.SharedPatternQPs(n=20, m=35, density=0.2, pattern_seed=5)
    elif case == "tiny":
        wl = R.workloads.SharedPatternQPs(n=1, m=1, density=1.0, pattern_seed=9)
    elif case == "unconstrained":
        wl = R.workloads.SharedPatternQPs(n=7, m=0, density=0.4, pattern_seed=3)
    else:
        wl = R.workloads.MPCStageQPs(N=6)
        perm = R.workloads.stage_permutation(*wl.dims)
    n, m = wl.n, wl.m
    P, q, A, l, u = wl.instance(0)
    pl = plan_export(wl.P_pattern, wl.A_pattern, perm=perm)
    sym = R.symbolic_analyze(wl.P_pattern, wl.A_pattern, perm=perm)
    assert pl["plan_ok"] == 1 and pl["nS"] >= sym["nnzL"]
    assert len(set(pl["LtoS"].tolist())) == sym["nnzL"] and (pl["LtoS"] < pl["nS"]).all() and (pl["LtoS"] >= 0).all()
    gs = pl["blob"][pl["po_gstart"]:pl["po_gstart"] + pl["ngroups"] + 1]
    assert gs[0] == 0 and gs[-1] == n + m and (np.diff(gs) > 0).all() and (np.diff(gs) <= 64).all()
    rho = np.full(m, 0.1)
    o = ob.OracleLinsys(P, A, 1e-6, rho, perm=sym["perm"])
    e = o.export()
    S = np.zeros(pl["nS"])
    S[pl["LtoS"]] = e["Lx"]
    rhs = np.random.default_rng(0).standard_normal(n + m)
    got = plan_emulate_solve(pl, S, e["Dinv"], rhs[sym["perm"]])
    ref = rhs[sym["perm"]].copy()
    ob.lib().orc_qdldl_solve.argtypes = [ob.c_int, ob.IP, ob.IP, ob.FP, ob.FP, ob.FP]
    ob.lib().orc_qdldl_solve(n + m, ob.ip(e["Lp"]), ob.ip(e["Li"]), ob.fp(e["Lx"]), ob.fp(e["Dinv"]), ob.fp(ref))
    assert np.max(np.abs(got - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("case", ["arrowhead", "small"])
def test_virtual_row_tables_cover_every_coupling_entry_once(case):
    """Arrowhead plans (csrc/rldl_plan.c, "virtual rows"): the coupling rows of the tail group are cut into pieces of
    at most T entries, one piece per lane.  Every out-of-group entry of L must appear exactly once, under its own row,
    with its own column and storage slot; T is minimal for 64 lanes; the pieces are sorted by length; and the entries are
    placed on the (padded) steps by an edge colouring: at every step the active lanes address different columns, so the
    LDS atomic adds of the backward pass never collide (whenever no column has more entries than there are steps)."""
    from osqp_recursive_ldl_amd.linsys import plan_export
    wl = R.workloads.SharedPatternQPs() if case == "arrowhead" else R.workloads.SharedPatternQPs(n=20, m=35, density=0.2, pattern_seed=5)
    pl = plan_export(wl.P_pattern, wl.A_pattern)
    sym = R.symbolic_analyze(wl.P_pattern, wl.A_pattern)
    if not pl["arrow_ok"]:
        pytest.skip("not an arrowhead plan")
    T, nv = pl["arrow_vsteps"], pl["arrow_vrows"]
    Tp = next(v for v in (12, 18, 24, (T + 1) & ~1) if T <= v)      # register bound the tables are padded to
    blob = pl["blob"].view(np.uint32)
    half = (Tp + 1) // 2
    vmap = blob[pl["po_avmap"]:pl["po_avmap"] + half * 64].reshape(half, 64)
    vcol = blob[pl["po_avcol"]:pl["po_avcol"] + half * 64].reshape(half, 64)
    vrow = pl["blob"][pl["po_avrow"]:pl["po_avrow"] + 64]
    gs = pl["blob"][pl["po_gstart"]:pl["po_gstart"] + pl["ngroups"] + 1]
    k = pl["arrow_group"]
    g0, g1 = int(gs[k]), int(gs[k + 1])
    # expected: all entries L(r, c) with r in the tail group and c outside it, keyed by storage slot
    want = {}
    Lp, Li = sym["Lp"], sym["Li"]
    for c in range(sym["N"] if "N" in sym else wl.n + wl.m):
        for p in range(Lp[c], Lp[c + 1]):
            r = int(Li[p])
            if g0 <= r < g1 and not (g0 <= c < g1):
                want[int(pl["LtoS"][p])] = (r, c)
    got, lens = {}, []
    cols_at_step = [[] for _ in range(Tp)]
    for lane in range(64):
        ln = 0
        for t in range(Tp):
            slot = int((vmap[t >> 1, lane] >> (16 * (t & 1))) & 0xffff)
            col = int((vcol[t >> 1, lane] >> (16 * (t & 1))) & 0xffff)
            if slot == 0xffff:
                assert col == 0
                continue
            assert lane < nv
            assert slot not in got
            got[slot] = (int(vrow[lane]), col)
            cols_at_step[t].append(col)
            ln += 1
        assert ln <= T
        lens.append(ln)
    col_deg = np.bincount([c for _, c in want.values()])
    if col_deg.max() <= Tp:
        assert all(len(c) == len(set(c)) for c in cols_at_step), "two lanes address the same column in one step"
    assert got == want and len(want) == pl["nO"]
    assert lens == sorted(lens, reverse=True) and all(v > 0 for v in lens[:nv]) and all(v == 0 for v in lens[nv:])
    rows = np.bincount([r - g0 for r, _ in want.values()], minlength=g1 - g0)
    pieces = lambda tt: int(sum((int(v) + tt - 1) // tt for v in rows))
    assert pieces(T) == nv <= 64 and (T == 1 or pieces(T - 1) > 64)


@pytest.mark.parametrize("case", ["arrowhead", "small", "g30"])
def test_tile_tables_reproduce_the_products_with_the_tail_inverse(case):
    """Tile plan of the arrowhead tail (csrc/rldl_plan.c, "tail inverse by register tiles"): the strictly lower part of
    Linv = L22^-1 is cut into ta x ta tiles, one per lane, rows rotated by the tile's block column and columns by its block
    row.  Emulates what a lane of k_tile_* does with the tables (po_tlane, po_tmap, po_tislot) on a random unit lower
    triangle: every strictly lower entry is stored exactly once, in (register, lane) order, and the forward / backward
    lane products followed by the atomic adds give Linv c and Linv' w."""
    from osqp_recursive_ldl_amd.linsys import plan_export
    wl = {"arrowhead": lambda: R.workloads.SharedPatternQPs(),
          "small": lambda: R.workloads.SharedPatternQPs(n=20, m=35, density=0.2, pattern_seed=5),
          "g30": lambda: R.workloads.SharedPatternQPs(n=30, m=60, density=0.2, pattern_seed=3)}[case]()
    pl = plan_export(wl.P_pattern, wl.A_pattern)
    if not pl["tile_ok"]:
        pytest.skip("no tile plan")
    gs = pl["blob"][pl["po_gstart"]:pl["po_gstart"] + pl["ngroups"] + 1]
    g = int(gs[pl["arrow_group"] + 1] - gs[pl["arrow_group"]])
    a, tq, nl, nTi = pl["tile_ta"], pl["tile_tq"], pl["tile_lanes"], pl["nTi"]
    assert a in (2, 3, 5, 7) and tq == -(-g // a) and nl == tq * (tq + 1) // 2 <= 64 and nTi == g * (g - 1) // 2
    blob = pl["blob"].view(np.uint32)
    tl = blob[pl["po_tlane"]:pl["po_tlane"] + 64]
    half = (a * a + 1) // 2
    tm = blob[pl["po_tmap"]:pl["po_tmap"] + half * 64].reshape(half, 64)
    ts = pl["blob"][pl["po_tislot"]:pl["po_tislot"] + g * 32].view(np.uint16).reshape(g, 64)
    rng = np.random.default_rng(11)
    L = np.tril(rng.standard_normal((g, g)), -1) * 0.3 + np.eye(g)
    Linv = np.linalg.inv(L)
    Ti = np.full(nTi, np.nan)
    for i in range(g):                                    # what k_tile_invert stores: lane c of row i -> slot
        for c in range(64):
            if ts[i, c] != 0xffff:
                assert c < i and np.isnan(Ti[ts[i, c]])
                Ti[ts[i, c]] = Linv[i, c]
    assert not np.isnan(Ti).any()
    slot = lambda k, lane: int((tm[k >> 1, lane] >> (16 * (k & 1))) & 0xffff)
    # slots are handed out in (register, lane) order: the loads of one register are one contiguous run
    seq = [slot(k, lane) for k in range(a * a) for lane in range(64) if slot(k, lane) != 0xffff]
    assert seq == list(range(nTi))
    cvec, wvec = rng.standard_normal(a * tq), rng.standard_normal(a * tq)
    cvec[g:] = 0.0; wvec[g:] = 0.0                        # padding rows of the last block row
    y, x = cvec.copy(), wvec.copy()
    for lane in range(64):
        if tl[lane] == 0xffffffff:
            assert lane >= nl and all(slot(k, lane) == 0xffff for k in range(a * a))
            continue
        I, J = int(tl[lane] & 0xff), int((tl[lane] >> 8) & 0xff)
        assert J <= I < tq
        T = np.array([[Ti[slot(s * a + u, lane)] if slot(s * a + u, lane) != 0xffff else 0.0 for u in range(a)] for s in range(a)])
        rows = [a * I + (s + J) % a for s in range(a)]
        cols = [a * J + (u + I) % a for u in range(a)]
        for s in range(a):
            for u in range(a):
                if rows[s] < g and cols[u] < rows[s]:
                    assert T[s, u] == Linv[rows[s], cols[u]]
                else:
                    assert T[s, u] == 0.0
        y[rows] += T @ cvec[cols]                         # tile_fwd: reads of c precede the adds
        x[cols] += T.T @ wvec[rows]                       # tile_bwd
    assert np.allclose(y[:g], Linv @ cvec[:g], rtol=1e-13, atol=1e-13)
    assert np.allclose(x[:g], Linv.T @ wvec[:g], rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("case", ["arrowhead", "small", "g30"])
def test_owner_gather_tables_hold_every_coupling_entry_under_its_column_owner(case):
    """ADMM slots and the owner gather of the tile kernels (csrc/rldl_plan.c): slot 0 holds the variables, slots 1-2 the
    constraints by decreasing column count; the lane that owns position c holds the entries L(r, c) of its column on the
    steps of its slot ([0, sp) for the first constraint slot, [sp, tk) for the second; sp = 3 tk / 4 or 2 tk / 3), each exactly once with its
    factor slot and its row (local to the tail group)."""
    from osqp_recursive_ldl_amd.linsys import plan_export
    wl = {"arrowhead": lambda: R.workloads.SharedPatternQPs(),
          "small": lambda: R.workloads.SharedPatternQPs(n=20, m=35, density=0.2, pattern_seed=5),
          "g30": lambda: R.workloads.SharedPatternQPs(n=30, m=60, density=0.2, pattern_seed=3)}[case]()
    pl = plan_export(wl.P_pattern, wl.A_pattern)
    sym = R.symbolic_analyze(wl.P_pattern, wl.A_pattern)
    if not pl["tile_admm_ok"]:
        pytest.skip("no ADMM slot plan")
    N, tk = wl.n + wl.m, pl["tile_tk"]
    assert pl["tile_vslots"] == 1 and pl["tile_slots"] <= 3 and tk in (16, 24, 32)
    ck = [pl["tile_ck0"], pl["tile_ck1"], pl["tile_ck2"]]
    sp = pl["tile_sp"]
    assert ck[0] == 0 and sp in (3 * tk // 4, (2 * tk // 3) & ~1) and ck[1] <= sp and ck[2] <= tk - sp
    tpos = pl["blob"][pl["po_tpos"]:pl["po_tpos"] + 192].reshape(3, 64)
    perm = sym["perm"]
    owned = tpos[tpos >= 0]
    assert sorted(owned.tolist()) == list(range(N))                                  # every permuted position has one owner
    assert all(perm[j] < wl.n for j in tpos[0][tpos[0] >= 0]) and all(perm[j] >= wl.n for j in tpos[1:][tpos[1:] >= 0])
    gs = pl["blob"][pl["po_gstart"]:pl["po_gstart"] + pl["ngroups"] + 1]
    g0, g1 = int(gs[pl["arrow_group"]]), int(gs[pl["arrow_group"] + 1])
    blob = pl["blob"].view(np.uint32)
    cm = blob[pl["po_cmap"]:pl["po_cmap"] + (tk // 2) * 64].reshape(tk // 2, 64)
    cr = blob[pl["po_crow"]:pl["po_crow"] + (tk // 2) * 64].reshape(tk // 2, 64)
    Lp, Li = sym["Lp"], sym["Li"]
    want = {}
    for c in range(N):
        for p in range(Lp[c], Lp[c + 1]):
            r = int(Li[p])
            if g0 <= r < g1 and not (g0 <= c < g1):
                want[int(pl["LtoS"][p])] = (r - g0, c)
    got = {}
    for t in (1, 2):
        lo, hi = (0, sp) if t == 1 else (sp, tk)
        for lane in range(64):
            for k in range(lo, hi):
                slot = int((cm[k >> 1, lane] >> (16 * (k & 1))) & 0xffff)
                row = int((cr[k >> 1, lane] >> (16 * (k & 1))) & 0xffff)
                if slot == 0xffff:
                    assert row == 0
                    continue
                assert tpos[t, lane] >= 0 and slot not in got and k - lo < ck[t]
                got[slot] = (row, int(tpos[t, lane]))
    assert got == want and len(want) == pl["nO"]
    cnt = np.bincount([c for _, c in want.values()], minlength=N)
    for t in (1, 2):                                                                   # decreasing column count inside a kind
        cols = tpos[1:].ravel()[tpos[1:].ravel() >= 0]
        assert all(cnt[cols[i]] >= cnt[cols[i + 1]] for i in range(len(cols) - 1))


def test_pattern_bucketing_groups_equal_patterns_in_order_of_first_appearance():
    """osqp_groups_bucket (host C): problems with equal (P pattern, A pattern) share a bucket, buckets are numbered by first
    appearance -- against a Python dict over the index arrays; patterns that differ in one row index or only in A are told apart."""
    from osqp_recursive_ldl_amd.groups import bucket_by_pattern
    from osqp_recursive_ldl_amd.linsys import CscPattern
    rng = np.random.default_rng(5)
    n, m = 12, 17
    base = []
    for k in range(6):
        P = sparse.random(n, n, density=0.25, random_state=100 + k, format="csc"); P = sparse.triu(P + P.T + sparse.eye(n), format="csc")
        A = sparse.random(m, n, density=0.3, random_state=200 + k, format="csc")
        base.append((P, A))
    base.append((base[0][0], base[1][1]))                        # same P as pattern 0, A of pattern 1: a bucket of its own
    A_mod = base[2][1].tolil(); zr, zc = np.nonzero(A_mod.toarray() == 0); A_mod[zr[0], zc[0]] = 1.0   # one more entry than pattern 2
    base.append((base[2][0], sparse.csc_matrix(A_mod)))
    order = rng.integers(0, len(base), size=60)
    Ps = [CscPattern(base[k][0]) for k in order]; As = [CscPattern(base[k][1]) for k in order]
    # values must not matter
    Ps[3] = Ps[3].with_values(rng.standard_normal(Ps[3].nnz))
    group, nb = bucket_by_pattern(Ps, As)
    seen = {}
    want = []
    for p, a in zip(Ps, As):
        key = (p.shape, a.shape, p.p.tobytes(), p.i.tobytes(), a.p.tobytes(), a.i.tobytes())
        want.append(seen.setdefault(key, len(seen)))
    assert nb == len(seen) and list(group) == want
    assert nb == len(set(order.tolist()))
    assert bucket_by_pattern([], [])[1] == 0

#!/usr/bin/env python3
"""Regenerate tests/golden/*.json from the reference's own Python test-data generators.

Run in the BUILD container only (needs /root/reference):

    python tests/golden/make_golden.py

How it works: the reference's tests/<suite>/generate_problem.py scripts (numpy/scipy only) build a
problem plus known answers and hand them to `utils.codegen_utils.generate_problem_data` /
`generate_data`, which print a C header.  We import those generator scripts unchanged, but put a
capturing module in `sys.modules['utils.codegen_utils']` first, so instead of a header we receive
the Python objects and dump them as JSON (sparse matrices as CSC triplets of lists).  Nothing of the
reference's source text is stored: the fixtures are data only.

Suites (reference file -> fixture):
    tests/solve_linsys/generate_problem.py              -> solve_linsys.json
    tests/update_matrices/generate_problem.py           -> update_matrices.json
    tests/basic_qp/generate_problem.py                  -> basic_qp.json
    tests/basic_qp2/generate_problem.py                 -> basic_qp2.json
    tests/non_cvx/generate_problem.py                   -> non_cvx.json
    tests/unconstrained/generate_problem.py             -> unconstrained.json
    tests/primal_dual_infeasibility/generate_problem.py -> primal_dual_infeasibility.json
    tests/lin_alg/generate_problem.py                   -> lin_alg.json
    tests/primal_infeasibility/generate_problem.py      -> primal_infeasibility.json, RE-SEEDED: the script draws q, l, u
        with `scipy.randn` (:17, :19-20), which current scipy no longer has (ordinary AttributeError) and which drew from
        numpy's unseeded global state anyway (the reference's own data.h is not reproducible).  The generator is imported
        unchanged with `scipy.randn` bound to numpy's Generator(PCG64(20261004)).standard_normal, so P, A and the
        infeasible row pair come from the script's own PCG64(2) stream and q, l, u from that recorded seed.
"""
import importlib
import json
import os
import sys
import types

import numpy as np
from scipy import sparse

REF_TESTS = "/root/reference/tests"
OUT = os.path.dirname(os.path.abspath(__file__))

_captured = {}


def _enc(v):
    if sparse.issparse(v):
        c = sparse.csc_matrix(v)
        c.sort_indices()
        return {"__csc__": True, "m": int(c.shape[0]), "n": int(c.shape[1]),
                "p": c.indptr.astype(int).tolist(), "i": c.indices.astype(int).tolist(),
                "x": [float(t) for t in c.data]}
    if isinstance(v, np.ndarray):
        return {"__vec__": True, "x": [_num(t) for t in v.ravel().tolist()]}
    if isinstance(v, (np.floating, float)):
        return _num(float(v))
    if isinstance(v, (np.integer, int)):
        return int(v)
    if isinstance(v, str):
        return v
    raise TypeError(type(v))


def _num(t):
    t = float(t)
    if np.isinf(t):
        return "inf" if t > 0 else "-inf"
    return t


def _generate_problem_data(P, q, A, l, u, problem_name, sols_data={}):
    d = {"P": _enc(P), "q": _enc(np.asarray(q, float)), "A": _enc(A),
         "l": _enc(np.asarray(l, float)), "u": _enc(np.asarray(u, float)),
         "n": int(P.shape[0]), "m": int(A.shape[0]),
         "sols": {k: _enc(v) for k, v in sols_data.items()}}
    _captured[problem_name] = d


def _generate_data(problem_name, sols_data):
    _captured[problem_name] = {"data": {k: _enc(v) for k, v in sols_data.items()}}


def main():
    utils_pkg = types.ModuleType("utils")
    utils_pkg.__path__ = []
    cu = types.ModuleType("utils.codegen_utils")
    cu.generate_problem_data = _generate_problem_data
    cu.generate_data = _generate_data
    utils_pkg.codegen_utils = cu
    sys.modules["utils"] = utils_pkg
    sys.modules["utils.codegen_utils"] = cu
    sys.path.insert(0, REF_TESTS)
    sys.dont_write_bytecode = True
    suites = ["solve_linsys", "update_matrices", "basic_qp", "basic_qp2", "non_cvx",
              "unconstrained", "primal_dual_infeasibility", "lin_alg"]
    import scipy
    if not hasattr(scipy, "randn"):                          # removed from the scipy namespace; see the module docstring
        _rs = np.random.Generator(np.random.PCG64(20261004))
        scipy.randn = lambda *shape: _rs.standard_normal(shape if len(shape) != 1 else shape[0])
    suites.append("primal_infeasibility")
    for s in suites:
        importlib.import_module(s + ".generate_problem")
        if s not in _captured:
            raise SystemExit("generator %s captured nothing" % s)
        with open(os.path.join(OUT, s + ".json"), "w") as f:
            json.dump(_captured[s], f, sort_keys=True)
        print("wrote", s + ".json")


if __name__ == "__main__":
    main()

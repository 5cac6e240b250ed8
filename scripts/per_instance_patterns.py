#!/usr/bin/env python3
"""BASELINE config 2, literal reading: batch = 1024 random sparse QPs (n=50, m=100, density 0.15), EVERY instance with its own sparsity
pattern (the reference's semantics: each osqp_setup analyses its own pattern, qdldl_interface.c:99-166).  One workspace per pattern
(host symbolic analysis per instance), one launch chain over all of them (osqp_multi_*): per-instance index tables are read from HBM /
L2 by the wave that owns the instance.  Prints setup time, solve time (factor + 200 ADMM iterations) and checks a sample against the
CPU oracle.  usage: per_instance_patterns.py [count]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from scipy import sparse
import osqp_recursive_ldl_amd as R

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=200, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
t0 = time.perf_counter()
problems = [R.workloads.SharedPatternQPs(pattern_seed=5000 + s).instance(0) for s in range(G)]
t1 = time.perf_counter()
g = R.OSQPBatchGroups(problems, **kw)
torch.cuda.synchronize()
t2 = time.perf_counter()
assert g.n_patterns == G
r = g.solve()
torch.cuda.synchronize()
def timed(fn, reps=21):
    """median of `reps` calls, each waited for (thousands of ctypes objects are alive here: an automatic full garbage collection
    in the middle of a short loop costs 70 ms once and used to land in a 5-call average)"""
    import gc
    fn(); torch.cuda.synchronize()
    gc.collect(); gc.disable()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
    gc.enable()
    return float(np.median(ts))
solve_ms = timed(g.solve)
vals = []
for idx, w in g.groups:
    i = int(idx[0])
    Pu = sparse.triu(sparse.csc_matrix(problems[i][0]), format="csc"); Pu.sort_indices()
    Ac = sparse.csc_matrix(problems[i][2]); Ac.sort_indices()
    vals.append((torch.from_numpy(Pu.data[None, :] * 1.02).cuda(), torch.from_numpy(Ac.data[None, :] * 0.99).cuda()))
def step():
    g.update_P_A(vals); g.solve()
step_ms = timed(step)
out = {"name": "config2_per_instance_patterns", "patterns": G, "instances": G, "n": 50, "m": 100, "one_launch_chain": g.one_launch,
       "generate_ms_per_instance": 1e3 * (t1 - t0) / G, "setup_ms_per_instance": 1e3 * (t2 - t1) / G, "setup_s": t2 - t1,
       "solve_ms_200_iterations": solve_ms, "qp_solves_per_sec_solve_only": G / (solve_ms * 1e-3),
       "step_ms_update_P_A_factor_200_iterations": step_ms, "qp_solves_per_sec": G / (step_ms * 1e-3)}
# algorithmic bytes of a tri-solve with per-instance int32 indices: 8 (nnzL + 3 N + m) + 4 nnzL + 4 (N + 1) + 4 N
nnzL = np.mean([w.linsys().dims()["nnzL"] for _, w in g.groups[:32]])
out["mean_nnzL"] = float(nnzL)
out["bytes_per_trisolve_per_instance_pattern"] = float(8 * (nnzL + 3 * 150 + 100) + 4 * nnzL + 4 * 151 + 4 * 150)
try:
    import oracle_bindings as ob
    g.update_P_A([(torch.from_numpy(np.ascontiguousarray(v[0].cpu().numpy() / 1.02)).cuda(), torch.from_numpy(np.ascontiguousarray(v[1].cpu().numpy() / 0.99)).cuda()) for v in vals])
    r = g.solve()
    worst = 0.0
    for k in range(0, G, max(1, G // 16)):
        P, q, A, l, u = problems[k]
        w = [w for idx, w in g.groups if int(idx[0]) == k][0]
        ro = ob.OracleOSQP(P, q, A, l, u, perm=w.linsys().export_symbolic()["perm"], **kw).solve()
        worst = max(worst, float(np.max(np.abs(r["x"][k].cpu().numpy() - ro["x"])) / max(1.0, np.max(np.abs(ro["x"])))))
    out["max_rel_err_vs_oracle_sample"] = worst
except Exception as e:                                             # (no oracle library next to the script)
    out["oracle_check"] = "skipped: %s" % e
print(json.dumps(out))
g.cleanup()

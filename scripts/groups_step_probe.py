#!/usr/bin/env python3
"""Pattern groups: a full step (new P / A values for every group, refactorisation, 200 iterations) -- how much of it is the
per-pattern update chains?  (diagnostic)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_recursive_ldl_amd as R
probs, wls = [], []
for s in range(8):
    wl = R.workloads.SharedPatternQPs(pattern_seed=2000 + s)
    wls.append(wl)
    probs += [wl.instance(b) for b in range(512)]
kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=200, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
g = R.OSQPBatchGroups(probs, **kw)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
vals = []
for (idx, w), wl in zip(g.groups, wls):
    Px, Ax, q, l, u = wl.values(512)
    vals.append((t(Px * 1.01), t(Ax * 0.99)))
def wall(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps
def updates():
    g.update_P_A(vals)
def updates_per_stream():
    for (idx, w), (Px, Ax) in zip(g.groups, vals): w.update_P_A(Px, Ax, wait=False)
def step():
    updates(); g.solve()
print(json.dumps(dict(one_launch=g.one_launch, updates_ms=wall(updates), updates_one_stream_per_pattern_ms=wall(updates_per_stream), solve_ms=wall(g.solve), step_ms=wall(step))))

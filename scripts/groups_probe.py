#!/usr/bin/env python3
"""Pattern groups: how well do the per-group solves overlap?  (diagnostic)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_recursive_ldl_amd as R
probs = []
for s in range(8):
    wl = R.workloads.SharedPatternQPs(pattern_seed=2000 + s)
    probs += [wl.instance(b) for b in range(512)]
kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=200, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
g = R.OSQPBatchGroups(probs, **kw)
def wall(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps
def launch_all():
    for _, w in g.groups: w.solve_async()
    for _, w in g.groups: w.wait(clone=False)
def one():
    w = g.groups[0][1]; w.solve_async(); w.wait(clone=False)
print(json.dumps(dict(hwq=os.environ.get("GPU_MAX_HW_QUEUES"), one_group_ms=wall(one), all_groups_kernels_ms=wall(launch_all), solve_ms=wall(g.solve),
                      loop_ms=[round(w.last_loop()[0], 3) for _, w in g.groups])))

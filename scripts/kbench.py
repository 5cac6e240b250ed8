#!/usr/bin/env python3
"""Kernel micro-benchmark for the fused ADMM-iteration kernel on the metric shape: a resident loop of KB_ITERS
iterations in one launch, and single-iteration launches (factor re-streamed every time) for comparison."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_recursive_ldl_amd as R
B = int(os.environ.get("KB_BATCH", "4096"))
wl = R.workloads.SharedPatternQPs()
Px, Ax, q, l, u = wl.values(B)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, t(Px), t(Ax), t(q), t(l), t(u), rho=0.1, max_iter=20, check_termination=0,
                adaptive_rho=0, warm_start=0, scaling=0)
w.solve()
ms = min(w.time_iteration(200) for _ in range(3))
d = w.linsys().dims()
by = 8 * (d["nnzL"] + 3 * 150 + 100) + 8 * (3 * 50 + 8 * 100)
K = int(os.environ.get("KB_ITERS", "200"))
w2 = R.OSQPBatch(wl.P_pattern, wl.A_pattern, t(Px), t(Ax), t(q), t(l), t(u), rho=0.1, max_iter=K, check_termination=0,
                 adaptive_rho=0, warm_start=0, scaling=0)
best = None
for _ in range(3):
    w2.solve()
    lm, it, gr = w2.last_loop()
    best = lm if best is None else min(best, lm)
print(json.dumps({"resident_loop_iters": it, "launches": gr, "loop_ms": best, "us_per_iteration": 1e3 * best / it,
                  "algorithmic_GBs": by * B * it / (best * 1e-3) / 1e9}))
print(json.dumps({"single_iteration_launches": True, "batch": B, "us_per_launch": 1e3 * ms, "GBs": by * B / (ms * 1e-3) / 1e9}))

#!/usr/bin/env python3
"""Default-settings solves (termination checks every 25 iterations, adaptive rho, scaling=10) on the metric shape:
the workload behind `metric_shape_batch4096_default_termination_scaling10` of bench_configs.py, for rocprofv3 traces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_recursive_ldl_amd as R
B = 4096
wl = R.workloads.SharedPatternQPs()
Px, Ax, q, l, u = wl.values(B)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, t(Px), t(Ax), t(q), t(l), t(u), rho=0.1, sigma=1e-6, alpha=1.6, max_iter=4000,
                check_termination=25, adaptive_rho=1, adaptive_rho_interval=100, eps_abs=1e-3, eps_rel=1e-3, warm_start=0, scaling=10)
for _ in range(3):
    w.solve()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    r = w.solve()
torch.cuda.synchronize()
print("ms per solve", 1e3 * (time.perf_counter() - t0) / 10, "loop", w.last_loop(), "iters", float(r["iter"].double().mean()), int(r["iter"].max()))

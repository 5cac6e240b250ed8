#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: the bench configuration (B=4096, n=50, m=100), two solves of 200 fused
ADMM iterations each, nothing else on the iteration kernel.  Run as
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python3 scripts/pmc_run.py   (and again with WRITE_SIZE)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_recursive_ldl_amd as R
B = 4096
wl = R.workloads.SharedPatternQPs()
Px, Ax, q, l, u = wl.values(B)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, t(Px), t(Ax), t(q), t(l), t(u), rho=0.1, sigma=1e-6, alpha=1.6, max_iter=200,
                check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
for _ in range(2):
    w.solve()
torch.cuda.synchronize()
print("iterations per launch:", w.last_loop()[1] // w.last_loop()[2])

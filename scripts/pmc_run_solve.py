#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes of the plugin `solve` kernel: the bench configuration (B=4096, n=50, m=100),
50 back-to-back launches of the batched permuted tri-solve, nothing else on that kernel.  Run as
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python3 scripts/pmc_run_solve.py   (and again with WRITE_SIZE)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_recursive_ldl_amd as R
B = 4096
wl = R.workloads.SharedPatternQPs()
Px, Ax, q, l, u = wl.values(B)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
ls = R.BatchLinsys(wl.P_pattern, wl.A_pattern, t(Px), t(Ax), 1e-6, t(np.full((B, wl.m), 0.1)))
b = torch.randn((B, wl.n + wl.m), dtype=torch.float64, device="cuda")
print("ms per launch:", ls.time_solve(b, reps=50))

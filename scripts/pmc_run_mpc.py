#!/usr/bin/env python3
"""Workload for the rocprofv3 passes on the MPC shape (BASELINE config 3: N=20, nx=12, nu=4, ny=10, nt=12, B=4096):
three stage factorisations (k_stage_factor_r), twenty plugin solves (k_plan_solve<., 22>) and one ADMM solve of twenty
iterations (k_plan_admm_loop<., 22>).  Run under  rocprofv3 --kernel-trace --stats  or  --pmc <counters>."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_recursive_ldl_amd as R
B = 4096
wl = R.workloads.MPCStageQPs(N=20)
Px, Ax, q, l, u = wl.values(B)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
dPx, dAx = t(Px), t(Ax)
rho = t(np.full((B, wl.m), 0.1))
ls = R.BatchLinsys.recursive(wl.dims, wl.P_pattern, wl.A_pattern, dPx, dAx, 1e-6, rho)
for _ in range(3):
    ls.update_rho_vec(rho)
b = torch.randn((B, wl.n + wl.m), dtype=torch.float64, device="cuda")
print("solve ms per launch:", ls.time_solve(b, reps=20))
ls.free()
kw = dict(rho=0.1, sigma=1e-6, alpha=1.6, max_iter=20, check_termination=0, adaptive_rho=0, warm_start=0, scaling=0)
w = R.OSQPBatch.recursive(wl.dims, wl.Q0, wl.Qi, wl.QN, wl.A0, wl.Ai, wl.Aij, wl.AN, t(q), t(l), t(u), **kw)   # osqp_setup_recursive: stage kernels
w.update_recursive(0, dPx, dAx)
w.solve()
torch.cuda.synchronize()
print("loop:", w.last_loop())

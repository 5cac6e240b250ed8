#!/usr/bin/env python3
"""MPC shape (N=20): tri-solve kernel time against the batch size (latency- or throughput-bound?).  Diagnostic."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_recursive_ldl_amd as R
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
wl = R.workloads.MPCStageQPs(N=int(os.environ.get("MPC_N", "20")))
Px, Ax, q, l, u = wl.values(8)
for B in [int(x) for x in os.environ.get("MPC_B", "256,1024,2048,4096,8192").split(",")]:
    rep = (B + 7) // 8
    dPx, dAx = t(np.tile(Px, (rep, 1))[:B]), t(np.tile(Ax, (rep, 1))[:B])
    rho = t(np.full((B, wl.m), 0.1))
    ls = R.BatchLinsys.recursive(wl.dims, wl.P_pattern, wl.A_pattern, dPx, dAx, 1e-6, rho)
    b = torch.randn((B, wl.n + wl.m), dtype=torch.float64, device=dev)
    ms = ls.time_solve(b, reps=20)
    print(json.dumps(dict(batch=B, us_per_launch=1e3 * ms, waves_per_cu=B / 256.0)), flush=True)
    ls.free()

import os, sys, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import osqp_recursive_ldl_amd as R
wl = R.workloads.SharedPatternQPs()
out = {}
for B in (1024, 2048, 3072, 4096, 8192, 16384):
    Px, Ax, q, l, u = wl.values(min(B, 4096))
    reps = (B + 4095) // 4096
    Px = np.tile(Px, (reps, 1))[:B]; Ax = np.tile(Ax, (reps, 1))[:B]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ls = R.BatchLinsys(wl.P_pattern, wl.A_pattern, t(Px), t(Ax), 1e-6, t(np.full((B, wl.m), 0.1)))
    b = torch.randn((B, 150), dtype=torch.float64, device="cuda")
    ls.time_solve(b, reps=20)
    ms = min(ls.time_solve(b, reps=100) for _ in range(3))
    out[B] = round(1e3 * ms, 2)
    ls.free()
print(json.dumps(out))

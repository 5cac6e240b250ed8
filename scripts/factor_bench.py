#!/usr/bin/env python3
"""Numeric refactorisation (update_matrices) of the metric shape, B=4096: device time per call (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_recursive_ldl_amd as R
B = 4096
wl = R.workloads.SharedPatternQPs()
Px, Ax, q, l, u = wl.values(B)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
dPx, dAx = t(Px), t(Ax)
ls = R.BatchLinsys(wl.P_pattern, wl.A_pattern, dPx, dAx, 1e-6, t(np.full((B, wl.m), 0.1)))
for _ in range(3):
    ls.update_matrices(dPx, dAx)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ls.update_matrices(dPx, dAx)
e1.record(); torch.cuda.synchronize()
print("ms per update_matrices", e0.elapsed_time(e1) / 10, "env", os.environ.get("RLDL_NO_ARROW_FACTOR"))

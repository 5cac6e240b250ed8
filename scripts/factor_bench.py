#!/usr/bin/env python3
"""Numeric refactorisation (update_matrices = KKT assembly + factorisation + tail inverse) of the metric shape: device time per call
(HIP events) at the batch sizes given on the command line (default 4096).  usage: factor_bench.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_recursive_ldl_amd as R
wl = R.workloads.SharedPatternQPs()
for B in [int(x) for x in sys.argv[1:]] or [4096]:
    Px, Ax, q, l, u = wl.values(min(B, 4096))
    Px = np.tile(Px, ((B + 4095) // 4096, 1))[:B]; Ax = np.tile(Ax, ((B + 4095) // 4096, 1))[:B]
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
    print("B", B, "ms per update_matrices", e0.elapsed_time(e1) / 10)
    tr = ls.trace_factor()
    if tr is not None:
        tr = tr.astype(np.float64) * 0.01                          # 100 MHz ticks -> us
        names = ["values_in_workspace", "head_contributions", "tail_to_registers", "tail_elimination", "factor_row_stored", "triangle_packed", "tail_inverse_stored"]
        q = lambda v: [round(float(x), 2) for x in np.percentile(v, [0, 50, 100])]
        print("   span", round(float(tr[:, 7].max() - tr[:, 0].min()), 2), "us; wave lifetime [min,p50,max]", q(tr[:, 7] - tr[:, 0]),
              {n: q(tr[:, k + 1] - tr[:, k]) for k, n in enumerate(names)})
    ls.free()

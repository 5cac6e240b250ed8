#!/usr/bin/env python3
"""Wave timeline of the fused ADMM-iteration kernel on the metric shape (osqp_batch_trace_iteration): where a wave's
time goes inside one iteration.  TRACE_ITERS=1: the single-iteration launch (loads included); TRACE_ITERS>1: the
last iteration of a resident multi-iteration launch (steady state).  Prints one JSON object."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import osqp_recursive_ldl_amd as R

B = int(os.environ.get("KB_BATCH", "4096"))
K = int(os.environ.get("TRACE_ITERS", "4"))
wl = R.workloads.SharedPatternQPs()
Px, Ax, q, l, u = wl.values(B)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, t(Px), t(Ax), t(q), t(l), t(u), rho=0.1, max_iter=20, check_termination=0,
                adaptive_rho=0, warm_start=0, scaling=0)
w.solve()
tr = None
for _ in range(3):
    tr = w.trace_iteration(K)
tk = 0.01                                            # us per tick (100 MHz)
names = ["rhs", "gather", "fwd_sweep", "bwd_sweep", "scatter", "update"]
ph = {nm: float(np.mean((tr[:, k + 1] - tr[:, k]) * tk)) for k, nm in enumerate(names)}
p90 = {nm: float(np.percentile((tr[:, k + 1] - tr[:, k]) * tk, 90)) for k, nm in enumerate(names)}
t0 = tr[:, 7].min()
out = dict(batch=B, iters=K, mean_phase_us=ph, p90_phase_us=p90, iteration_us=float(np.mean(tr[:, 6] - tr[:, 0]) * tk),
           load_phase_us=float(np.mean(tr[:, 0] - tr[:, 7]) * tk) if K == 1 else None,
           wave_span_us=float((tr[:, 6].max() - t0) * tk), start_spread_us=float((tr[:, 7].max() - t0) * tk))
if os.environ.get("TRACE_DUMP"):
    np.save(os.environ["TRACE_DUMP"], tr)
print(json.dumps(out))

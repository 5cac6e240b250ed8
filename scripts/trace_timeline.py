#!/usr/bin/env python3
"""Wave timeline of the fused ADMM-iteration kernel on the metric shape (osqp_batch_trace_iteration):
where a wave's lifetime goes and how the 4096 waves are spread over the CUs in time.  Prints one JSON object."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import osqp_recursive_ldl_amd as R

B = int(os.environ.get("KB_BATCH", "4096"))
wl = R.workloads.SharedPatternQPs()
Px, Ax, q, l, u = wl.values(B)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
w = R.OSQPBatch(wl.P_pattern, wl.A_pattern, t(Px), t(Ax), t(q), t(l), t(u), rho=0.1, max_iter=20, check_termination=0,
                adaptive_rho=0, warm_start=0, scaling=0)
w.solve()
ms = min(w.time_iteration(200) for _ in range(3))
tr = None
for _ in range(3):
    tr = w.trace_iteration()
tk = 0.01                                            # us per tick (100 MHz)
t0 = tr[:, 0].min()
start = (tr[:, 0] - t0) * tk
end = (tr[:, 6] - t0) * tk
names = ["inputs_arrive", "rhs_built", "gather+triangle_arrive", "sweeps", "scatter", "epilogue"]
ph = {nm: float(np.mean((tr[:, k + 1] - tr[:, k]) * tk)) for k, nm in enumerate(names)}
ph_p90 = {nm: float(np.percentile((tr[:, k + 1] - tr[:, k]) * tk, 90)) for k, nm in enumerate(names)}
life = end - start
hist, edges = np.histogram(start, bins=12)
cu = tr[:, 7] & 0xffff
sweep_cycles = tr[:, 7] >> 16
out = dict(batch=B, us_per_launch=1e3 * ms, span_us=float(end.max()), mean_phase_us=ph, p90_phase_us=ph_p90,
           wave_life_us=dict(mean=float(life.mean()), p10=float(np.percentile(life, 10)), p90=float(np.percentile(life, 90))),
           start_hist=dict(counts=hist.tolist(), edges_us=[round(float(e), 2) for e in edges]),
           late_starts=int((start > 0.25 * end.max()).sum()), distinct_cu_ids=int(len(np.unique(cu))),
           waves_per_cu=dict(min=int(np.bincount(cu - cu.min()).min()), max=int(np.bincount(cu - cu.min()).max())),
           end_hist=np.histogram(end, bins=12)[0].tolist(),
           sweep_cycles_mean=float(sweep_cycles.mean()), sweep_clock_GHz=float(sweep_cycles.mean() / (np.mean(tr[:, 4] - tr[:, 3]) * 10.0)))
if os.environ.get("TRACE_DUMP"):
    np.save(os.environ["TRACE_DUMP"], tr)
print(json.dumps(out))

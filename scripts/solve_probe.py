#!/usr/bin/env python3
"""Probe of the plugin `solve` kernel on the metric shape (n=50, m=100): launch duration at several batch sizes (same handle
back to back = factor rows cached in the Infinity Cache between launches; rotation of handles = every launch streams from HBM),
the wave timeline of one launch (rldl_batch_trace_solve) and, with --save, the solution of a fixed right-hand side for a
bit-level comparison between kernel versions (RLDL_SOLVE_V2=1 selects the round-2 kernel).
usage: solve_probe.py [--save out.npy] [--batches 1024,4096,...] [--rot 5] [--json out.json]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import osqp_recursive_ldl_amd as R

ap = argparse.ArgumentParser()
ap.add_argument("--save"); ap.add_argument("--json")
ap.add_argument("--batches", default="4096")
ap.add_argument("--rot", type=int, default=5)
a = ap.parse_args()
wl = R.workloads.SharedPatternQPs()
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
out = {"kernel_version": "round-2 (RLDL_SOLVE_V2)" if os.environ.get("RLDL_SOLVE_V2") else "round-3", "batches": {}}


def handle(B, seed0=0):
    Px, Ax, q, l, u = wl.values(min(B, 4096), seed0=seed0)
    reps = (B + 4095) // 4096
    Px = np.tile(Px, (reps, 1))[:B]; Ax = np.tile(Ax, (reps, 1))[:B]
    return R.BatchLinsys(wl.P_pattern, wl.A_pattern, t(Px), t(Ax), 1e-6, t(np.full((B, wl.m), 0.1)))


for B in [int(x) for x in a.batches.split(",")]:
    ls = handle(B)
    d = ls.dims()
    tri = 8 * (d["nnzL"] + 3 * (wl.n + wl.m) + wl.m)
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    rhs = torch.randn((B, 150), dtype=torch.float64, device="cuda", generator=g)
    if a.save and B == 4096:
        np.save(a.save, ls.solve(rhs.clone()).cpu().numpy())
    b = rhs.clone()
    ls.time_solve(b, reps=20)
    ms = min(ls.time_solve(b, reps=200) for _ in range(3))
    rec = {"resident_us": 1e3 * ms, "resident_frac_of_8TBs": tri * B / (ms * 1e-3) / 8e12}
    if a.rot > 1 and B <= 8192:
        hs = [ls] + [handle(B, seed0=1000 * k) for k in range(1, a.rot)]
        bs = [b] + [rhs.clone() for _ in range(1, a.rot)]
        R.BatchLinsys.time_solve_rotating(hs, bs, reps=4 * a.rot)
        msr = min(R.BatchLinsys.time_solve_rotating(hs, bs, reps=40 * a.rot) for _ in range(3))
        rec.update(rotating_us=1e3 * msr, rotating_frac_of_8TBs=tri * B / (msr * 1e-3) / 8e12, rotation=a.rot,
                   rotation_working_set_MB=a.rot * (tri * B) / 1e6)
        for h in hs:
            h.set_cache_policy("stream")                           # non-temporal loads of the factor rows (rldl_batch_set_cache_policy)
        R.BatchLinsys.time_solve_rotating(hs, bs, reps=4 * a.rot)
        msn = min(R.BatchLinsys.time_solve_rotating(hs, bs, reps=40 * a.rot) for _ in range(3))
        b2 = rhs.clone()
        ls.time_solve(b2, reps=20)
        msrn = min(ls.time_solve(b2, reps=200) for _ in range(3))
        rec.update(rotating_stream_policy_us=1e3 * msn, rotating_stream_policy_frac_of_8TBs=tri * B / (msn * 1e-3) / 8e12,
                   resident_stream_policy_us=1e3 * msrn)
        ls.set_cache_policy("auto")
        for h in hs[1:]:
            h.free()
    tr = ls.trace_solve(rhs.clone())
    if tr is not None:
        tr = tr.astype(np.float64) * 0.01                          # 100 MHz ticks -> us
        t00 = tr[:, 0].min()
        q = lambda v: [round(float(x), 2) for x in np.percentile(v, [0, 10, 50, 90, 100])]
        rec["timeline_us"] = {"span_first_start_to_last_end": round(float(tr[:, 6].max() - t00), 2),
                              "wave_start_after_first [min,p10,p50,p90,max]": q(tr[:, 0] - t00),
                              "loads_issued_after_start": q(tr[:, 7] - tr[:, 0]),
                              "loads_landed_after_start": q(tr[:, 1] - tr[:, 0]),
                              "forward_gather": q(tr[:, 2] - tr[:, 1]), "forward_product": q(tr[:, 3] - tr[:, 2]),
                              "backward_product": q(tr[:, 4] - tr[:, 3]), "scatter": q(tr[:, 5] - tr[:, 4]),
                              "epilogue_stores_issued": q(tr[:, 6] - tr[:, 5]),
                              "wave_lifetime": q(tr[:, 6] - tr[:, 0]),
                              "wave_end_after_first_start": q(tr[:, 6] - t00)}
    out["batches"][B] = rec
    ls.free()
print(json.dumps(out))
if a.json:
    json.dump(out, open(a.json, "w"), indent=1)

#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc passes to the JSON summaries bench.py reads from profiles/.
  pmc_reduce.py fused  <out.json> <dir> [<dir> ...]   SQ counters of the fused iteration kernel (scripts/pmc_run.py), per wave-iteration
  pmc_reduce.py traffic <out.json> <fetch_dir> <write_dir> <kernel substring> <algorithmic bytes per launch> <batch>
  pmc_reduce.py mpc   <out.json> <dir> [<dir> ...]   SQ counters of the MPC kernels (scripts/pmc_run_mpc.py), per wave and launch
"""
import csv, glob, json, os, sys


def collect(dirs, kern):
    acc = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if kern in row["Kernel_Name"]:
                    acc.setdefault(row["Counter_Name"], {}).setdefault((f, row["Dispatch_Id"]), 0.0)
                    acc[row["Counter_Name"]][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
    return {c: (sum(v.values()) / len(v), len(v)) for c, v in acc.items()}


mode, out = sys.argv[1], sys.argv[2]
if mode == "fused":
    B, iters = 4096, 200
    kern = next(k for k in ("k_tile_admm", "k_arrow_admm", "k_plan_admm") if collect(sys.argv[3:], k))
    c = collect(sys.argv[3:], kern)
    res = {"kernel": kern, "batch": B, "iterations_per_launch": iters, "workload": "scripts/pmc_run.py (bench configuration, two solves of 200 fused iterations)",
           "launches": {k: v[1] for k, v in c.items()}, "per_launch": {k: v[0] for k, v in c.items()},
           "per_wave_iteration": {k: v[0] / (B * iters) for k, v in c.items()},
           "units": "SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles; SQ_LDS_IDX_ACTIVE and SQ_LDS_BANK_CONFLICT count LDS-array cycles; SQ_INSTS_* count wave instructions"}
elif mode == "mpc":
    B = 4096
    res = {"workload": "scripts/pmc_run_mpc.py (MPC shape N=20 nx=12 nu=4 ny=10 nt=12, batch 4096)", "batch": B,
           "units": "per wave (= per instance) and launch; SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles; SQ_LDS_* count LDS-array cycles"}
    for kern in ("k_stage_factor_r", "k_stage_invert", "k_plan_solve", "k_plan_admm_loop"):
        c = collect(sys.argv[3:], kern)
        res[kern] = {"launches": {k: v[1] for k, v in c.items()}, "per_wave_and_launch": {k: v[0] / B for k, v in c.items()}}
    # HBM traffic when the FETCH_SIZE / WRITE_SIZE passes are among the directories (KB per launch; gfx950 correction: reads x 2)
    for kern in ("k_stage_factor_r", "k_stage_invert", "k_plan_solve", "k_plan_admm_loop"):
        c = collect(sys.argv[3:], kern)
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            res[kern]["traffic_bytes_per_launch"] = 2 * 1024 * c["FETCH_SIZE"][0] + 1024 * c["WRITE_SIZE"][0]
else:
    fd, wd, kern, alg, B = sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
    fe, wr = collect([fd], kern).get("FETCH_SIZE", (0.0, 0)), collect([wd], kern).get("WRITE_SIZE", (0.0, 0))
    res = {"kernel": kern, "batch": B, "FETCH_SIZE": {"launches": fe[1], "avg_KB_per_launch": fe[0]}, "WRITE_SIZE": {"launches": wr[1], "avg_KB_per_launch": wr[0]},
           "correction": "gfx950: FETCH_SIZE counts 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> read bytes = 2 x FETCH_SIZE x 1024; "
                         "WRITE_SIZE x 1024 taken as is; Infinity-Cache hits are counted, not excluded",
           "traffic_bytes_per_launch": 2 * 1024 * fe[0] + 1024 * wr[0], "algorithmic_bytes_per_launch": alg}
    res["traffic_over_algorithmic"] = res["traffic_bytes_per_launch"] / alg
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res)[:600])

#!/usr/bin/env python3
"""Reduce a rocprofv3 --pmc pass of SQ counters to per-wave-iteration figures for k_arrow_admm (scripts/pmc_run.py)."""
import csv, glob, json, os, sys
d = sys.argv[1]; kern = sys.argv[2] if len(sys.argv) > 2 else "k_arrow_admm"
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
out = {}
for c, v in acc.items():
    vals = list(v.values())
    out[c] = sum(vals) / len(vals)
per = 4096 * 200.0
print(json.dumps({"per_launch": out, "per_wave_iteration": {k: v / per for k, v in out.items()}}))

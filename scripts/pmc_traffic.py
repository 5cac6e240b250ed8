#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of scripts/pmc_run.py into profiles/<name>.json.
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [kernel substring]"""
import csv, glob, json, os, sys

def collect(d, counter, kern):
    vals = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kern in row["Kernel_Name"] and row["Counter_Name"] == counter:
                key = (f, row["Dispatch_Id"])
                vals[key] = vals.get(key, 0.0) + float(row["Counter_Value"])
    return list(vals.values())

fd, wd, out = sys.argv[1:4]
kern = sys.argv[4] if len(sys.argv) > 4 else "k_arrow_admm"
fe, wr = collect(fd, "FETCH_SIZE", kern), collect(wd, "WRITE_SIZE", kern)
B, iters, per_iter = 4096, 200, 27704
res = {
    "FETCH_SIZE": {"launches": len(fe), "avg_KB_per_launch": sum(fe) / max(len(fe), 1)},
    "WRITE_SIZE": {"launches": len(wr), "avg_KB_per_launch": sum(wr) / max(len(wr), 1)},
    "kernel": "%s, batch %d, n=50 m=100, %d fused iterations per launch (scripts/pmc_run.py)" % (kern, B, iters),
    "correction": "gfx950: FETCH_SIZE counts 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> read bytes = "
                  "2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 taken as is",
}
res["traffic_bytes_per_launch"] = 2 * 1024 * res["FETCH_SIZE"]["avg_KB_per_launch"] + 1024 * res["WRITE_SIZE"]["avg_KB_per_launch"]
res["algorithmic_bytes_per_launch"] = per_iter * B * iters
res["traffic_over_algorithmic"] = res["traffic_bytes_per_launch"] / res["algorithmic_bytes_per_launch"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))

#!/usr/bin/env python3
"""Average per-dispatch counter values of the kernels matching argv[2] in a rocprofv3 --pmc output directory argv[1];
argv[3] (optional) = waves per dispatch to divide by."""
import csv, glob, json, os, sys
d, kern = sys.argv[1], sys.argv[2]
per = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
print(json.dumps({c: sum(v.values()) / len(v) / per for c, v in acc.items()}))

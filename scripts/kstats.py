#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as 'calls avg_us total_ms name' (names cut), largest total first.  usage: kstats.py <csv> [rows]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    name = re.sub(r"void \(anonymous namespace\)::|\(anonymous namespace\)::", "", r["Name"]).split("(")[0][:60]
    print("%6s %10.2f us %9.3f ms  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, name))

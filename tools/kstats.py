#!/usr/bin/env python3
"""Print selected rows of a rocprofv3 kernel_stats.csv: kstats.py <dir or csv> <regex> [steps in the run]"""
import csv, glob, os, re, sys
path, pat = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
f = path if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if re.search(pat, n):
        short = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
        ms = float(r["TotalDurationNs"]) / 1e6 / steps
        tot += ms
        print(f"{short:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:9.1f} us  per step {ms:8.2f} ms")
print(f"total per step {tot:.2f} ms")

#!/usr/bin/env python3
"""Per kernel VARIANT: HBM GB per step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, corrections of tools/pmc_traffic.py), time per step from a kernel
trace of the same command, and the GB/s that implies - which launches of a step run at the HBM roof and which carry bytes they should not.
    pmc_traffic_variants.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel_trace.csv> <steps in the PMC run> <steps in the trace> [top = 40]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"::(\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name.split("(")[0][-60:]


def main():
    fcsv, wcsv, trace, psteps, tsteps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    top = int(sys.argv[6]) if len(sys.argv) > 6 else 40
    f, w, ns, n = defaultdict(float), defaultdict(float), defaultdict(float), defaultdict(int)
    for path, counter, acc in ((fcsv, "FETCH_SIZE", f), (wcsv, "WRITE_SIZE", w)):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter:
                    acc[short(row["Kernel_Name"])] += float(row["Counter_Value"])
    with open(trace, newline="") as fh:
        for row in csv.DictReader(fh):
            k = short(row["Kernel_Name"])
            ns[k] += float(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            n[k] += 1
    rows = []
    for k in set(f) | set(w):
        gb_r, gb_w = 2.0 * f[k] * 1024 / 1e9 / psteps, w[k] * 1024 / 1e9 / psteps
        ms = ns.get(k, 0.0) / 1e6 / tsteps
        rows.append((gb_r + gb_w, gb_r, gb_w, ms, n.get(k, 0) / tsteps, k))
    rows.sort(reverse=True)
    print(" GB/step   read  write  ms/step   n/step   TB/s  kernel variant")
    for tot, r, wv, ms, cnt, k in rows[:top]:
        print("%8.1f %6.1f %6.1f %8.2f %8.1f %6.2f  %s" % (tot, r, wv, ms, cnt, (tot / ms) if ms else 0.0, k[:120]))
    print("total %.1f GB/step over %d variants" % (sum(r[0] for r in rows), len(rows)))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Group the kernel trace of a graphed (or eager) step by (kernel, workgroups): launches and microseconds per step over the LAST `steps` steps of the trace.
    python tools/trace_by_grid.py kernel_trace.csv launches_per_step [steps = 3]
(launches_per_step: as printed by tools/kstats_steady.py for the same trace; the trace's tail is cut into whole steps of that many launches.)"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?", name)
    base = m.group(1) if m else name
    targs = (m.group(2) or "") if m else ""
    return (base + targs)[:110]


def main():
    path, per_step = sys.argv[1], int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    rows = [r for r in csv.DictReader(open(path)) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-per_step * steps:]
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows:
        wgs = 1
        for ax in "XYZ":
            wgs *= max(1, int(r[f"Grid_Size_{ax}"]) // max(1, int(r[f"Workgroup_Size_{ax}"])))
        k = (short(r["Kernel_Name"]), wgs)
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3 / steps
    total = sum(v[1] for v in agg.values()) / steps
    print(f"{len(rows) // steps} launches per step, {total:.0f} us of kernel time per step, {span:.0f} us from first start to last end per step")
    print(f"{'us/step':>9} {'n/step':>7} {'us each':>8} {'workgroups':>10}  kernel")
    small = 0.0
    for (name, wgs), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{us / steps:9.1f} {n / steps:7.1f} {us / n:8.1f} {wgs:10d}  {name}")
        if wgs <= 64:
            small += us / steps
    print(f"launches of <= 64 workgroups: {small:.0f} us per step")


if __name__ == "__main__":
    main()

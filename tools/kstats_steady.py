#!/usr/bin/env python3
"""Per-STEP kernel time of a rocprofv3 --kernel-trace run of bench.py (the optimizer kernel ends a step): all kernels and the conv implicit-GEMM family (forward / data
gradient / weight gradient kernels, their split-K reduce and the Winograd transforms - what bench.py's HIP-event classes conv_fwd / conv_dgrad / conv_wgrad cover).

Reconciles the two family times a reader meets: `rocprofv3 --stats` totals divided by the number of steps INCLUDE the first step of the process (cold: filter transposes, first
touches, one-off initialisation kernels), bench.py's `roofline.kernel_ms_per_step` is measured over steady-state steps after the timed region.
    python tools/kstats_steady.py <kernel_trace.csv> [optimizer kernel substring = sgd_nesterov|adamw_k]"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_classes import in_conv_family  # noqa: E402


def main():
    path = sys.argv[1]
    marks = tuple((sys.argv[2] if len(sys.argv) > 2 else "sgd_nesterov|adamw_k").split("|"))
    with open(path, newline="") as fh:
        rows = sorted(csv.DictReader(fh), key=lambda r: int(r["Start_Timestamp"]))
    steps, cur = [], [0.0, 0.0, 0]
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        cur[1] += d
        cur[2] += 1
        if in_conv_family(r["Kernel_Name"]):
            cur[0] += d
        if any(m in r["Kernel_Name"] for m in marks):
            steps.append(cur)
            cur = [0.0, 0.0, 0]
    for i, (fam, tot, n) in enumerate(steps):
        print(f"step {i}: conv family {fam:8.2f} ms   all kernels {tot:8.2f} ms   {n} launches" + ("   <- first step of the process (cold)" if i == 0 else ""))
    if len(steps) > 1:
        tail = steps[1:]
        print(f"all {len(steps)} steps / {len(steps)} (what --stats totals give): family {sum(s[0] for s in steps) / len(steps):.2f} ms, all {sum(s[1] for s in steps) / len(steps):.2f} ms")
        print(f"steady state (steps 1..): family {sum(s[0] for s in tail) / len(tail):.2f} ms, all {sum(s[1] for s in tail) / len(tail):.2f} ms")


if __name__ == "__main__":
    main()

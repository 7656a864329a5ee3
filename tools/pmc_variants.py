#!/usr/bin/env python3
"""Per kernel VARIANT (template instantiation) of one training step: time, launches, effective clock, matrix-pipe busy fraction and the wave stall buckets, from one
rocprofv3 pass that collected --kernel-trace and --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
(tools/exp/r04_variants.sh).  SQ_WAIT_ANY = parked at s_waitcnt / barrier, SQ_WAIT_INST_ANY = issue stall (matrix pipe busy, dependencies), fractions of SQ_WAVE_CYCLES.

    pmc_variants.py <counter_collection.csv> <kernel_trace.csv> <steps in the run> [rows = 40]"""
import collections
import csv
import re
import sys


def main():
    cc, tr, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    nrows = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    disp = {}
    for r in csv.DictReader(open(tr)):
        disp[r["Dispatch_Id"]] = (r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt, dur, seen = collections.Counter(), collections.defaultdict(float), set()
    for r in csv.DictReader(open(cc)):
        name, us = disp[r["Dispatch_Id"]]
        m = re.search(r"(\w+_k)(<[^>]*>)?", name)
        k = (m.group(1) + (m.group(2) or "")) if m else name[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            cnt[k] += 1
            dur[k] += us
    rows = []
    for k, a in agg.items():
        gui, wc = a["GRBM_GUI_ACTIVE"] / 8, a["SQ_WAVE_CYCLES"] or 1.0
        rows.append((dur[k] / steps / 1e3, k, cnt[k] / steps, gui / dur[k] / 1e3 if dur[k] else 0.0, a["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024) if gui else 0.0,
                     a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc, a["SQ_ACTIVE_INST_ANY"] / wc))
    rows.sort(reverse=True)
    print("%8s %6s %5s %5s %6s %6s %6s  kernel variant" % ("ms/step", "n/step", "GHz", "busy", "parked", "istall", "active"))
    for r in rows[:nrows]:
        print("%8.2f %6.1f %5.2f %5.2f %6.2f %6.2f %6.2f  %s" % (r[0], r[2], r[3], r[4], r[5], r[6], r[7], r[1][:120]))
    print("total %.2f ms/step over %d variants" % (sum(r[0] for r in rows), len(rows)))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass (--output-format csv) into matrix-pipe
utilisation per kernel class of one training step.

    pmc_mfma.py <counter_collection.csv> <steps in the run> <out.json> [algorithmic GFLOP per step of the conv family | -] [kernel_trace.csv of the same pass]

Per dispatch rocprofv3 reports the counters summed over their instances:
  * SQ_VALU_MFMA_BUSY_CYCLES - cycles the MFMA pipe of a SIMD is busy, summed over all 1024 SIMDs (256 CUs x 4).  One
    v_mfma_f32_32x32x2_f32 holds its SIMD's pipe for 64 cycles and does 4096 FLOP, so EXECUTED FLOP = 64 x busy cycles: the file also
    gives executed vs algorithmic FLOP (tile padding, the stem's 4th channel, dense ResNeXt groups show up here);
  * GRBM_GUI_ACTIVE - GPU-active cycles summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS section): kernel cycles = value / 8;
  * mfma_busy_frac = MFMA busy / (kernel cycles x 1024 SIMDs): the fraction of all matrix-pipe cycles of the chip that did MFMA work
    while the kernel ran (MfmaUtil of counter_defs.yaml); SQ_BUSY_CYCLES is kept as reported for reference;
  * with the pass's kernel trace: effective_clock_ghz = kernel cycles / kernel time per class - the clock the chip held under that load - and the wave stall
    buckets SQ_WAIT_ANY (parked at s_waitcnt / barrier), SQ_WAIT_INST_ANY (issue stall), SQ_ACTIVE_INST_ANY as fractions of SQ_WAVE_CYCLES when collected.
"""
import csv
import os
import json
import sys
from collections import defaultdict

from pmc_traffic import classify, build_identity

SIMDS, XCDS = 1024, 8


def main():
    path, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    algo_gflop = float(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] not in ("", "-") else None
    trace = sys.argv[5] if len(sys.argv) > 5 else None          # the same pass's kernel trace: durations -> effective clock per class
    per = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            c = classify(row["Kernel_Name"])
            per[c][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[c].add(row["Dispatch_Id"])
    dur_ns = defaultdict(float)
    if trace and os.path.exists(trace):
        with open(trace, newline="") as fh:
            for row in csv.DictReader(fh):
                dur_ns[classify(row["Kernel_Name"])] += float(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    res = {}
    for c, v in sorted(per.items()):
        busy, gui, sq = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0), v.get("SQ_BUSY_CYCLES", 0.0)
        cyc = gui / XCDS
        res[c] = {"launches": len(launches[c]) // steps, "mfma_busy_cycles_per_step": round(busy / steps), "kernel_cycles_per_step": round(cyc / steps),
                  "sq_busy_cycles_per_step": round(sq / steps), "mfma_busy_frac": round(busy / (cyc * SIMDS), 4) if cyc else None,
                  "executed_gflop_per_step": round(64.0 * busy / steps / 1e9, 1)}
        for name in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if v.get("SQ_WAVE_CYCLES") and name != "SQ_WAVE_CYCLES" and name in v:
                res[c][name.lower() + "_frac_of_wave_cycles"] = round(v[name] / v["SQ_WAVE_CYCLES"], 4)
        if dur_ns.get(c) and cyc:
            # cycles the chip counted while the class's kernels ran / the time they took: the clock the chip held under that load (DVFS; reads high for
            # dispatches much shorter than 0.3 ms - MI355X_MICROARCH.md) - the fp32 MFMA peak of 157.3 TFLOP/s is quoted at 2.4 GHz
            res[c]["kernel_ms_per_step_this_pass"] = round(dur_ns[c] / steps / 1e6, 3)
            res[c]["effective_clock_ghz"] = round(cyc / dur_ns[c], 3)
    conv = [res[k] for k in ("conv_fwd", "conv_dgrad", "conv_wgrad") if k in res]
    summary = {}
    if conv:
        busy = sum(r["mfma_busy_cycles_per_step"] for r in conv)
        cyc = sum(r["kernel_cycles_per_step"] for r in conv)
        summary = {"conv_family_mfma_busy_frac": round(busy / (cyc * SIMDS), 4), "conv_family_executed_gflop_per_step": round(64.0 * busy / 1e9, 1)}
        if algo_gflop:
            summary["conv_family_algorithmic_gflop_per_step"] = algo_gflop
            summary["executed_over_algorithmic"] = round(64.0 * busy / 1e9 / algo_gflop, 3)
        if all("effective_clock_ghz" in r for r in conv):
            ms = sum(r["kernel_ms_per_step_this_pass"] for r in conv)
            summary["conv_family_effective_clock_ghz"] = round(cyc / (ms * 1e6), 3)
            summary["conv_family_ms_per_step_this_pass"] = round(ms, 2)
        total_cyc = sum(r["kernel_cycles_per_step"] for r in res.values())
        summary["whole_step_mfma_busy_frac"] = round(sum(r["mfma_busy_cycles_per_step"] for r in res.values()) / (total_cyc * SIMDS), 4)
    json.dump({"normalisation": "MFMA busy cycles (sum over SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); executed FLOP = 64 x busy cycles (fp32 32x32x2: 64 cycles, 4096 FLOP)",
               "steps_in_run": steps, **build_identity(), "summary": summary, "per_class": res}, open(out, "w"), indent=1)
    print(json.dumps({"summary": summary, "per_class": res}, indent=1))


if __name__ == "__main__":
    main()

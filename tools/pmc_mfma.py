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
from kernel_classes import is_bf16x3

SIMDS, XCDS = 1024, 8
# FLOP per MFMA-busy cycle of one SIMD: v_mfma_f32_32x32x2_f32 = 4096 FLOP in 64 cycles; v_mfma_f32_16x16x32_bf16 = 16384 FLOP in 16 cycles.  A bf16x3 kernel
# (kernel_classes.is_bf16x3) spends SIX bf16 piece products per fp32 product: its fp32-equivalent executed FLOP = bf16 FLOP / 6.
FLOP_PER_BUSY_F32, FLOP_PER_BUSY_BF16, TERMS = 64.0, 1024.0, 6.0


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
            if row["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":        # split the busy cycles by the instruction the kernel issues
                per[c]["_busy_bf16" if is_bf16x3(row["Kernel_Name"]) else "_busy_f32"] += float(row["Counter_Value"])
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
                  # fp32-product equivalents: fp32-MFMA cycles x 64 + bf16-piece cycles x 1024 / 6
                  "executed_gflop_per_step": round((FLOP_PER_BUSY_F32 * v.get("_busy_f32", 0.0) + FLOP_PER_BUSY_BF16 / TERMS * v.get("_busy_bf16", 0.0)) / steps / 1e9, 1),
                  "bf16_piece_busy_share": round(v.get("_busy_bf16", 0.0) / busy, 4) if busy else None,
                  "bf16_pflops_while_running": None}
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
        ex = sum(r["executed_gflop_per_step"] for r in conv)
        summary = {"conv_family_mfma_busy_frac": round(busy / (cyc * SIMDS), 4), "conv_family_executed_gflop_per_step": round(ex, 1),
                   "conv_family_bf16_piece_busy_share": round(sum((r["bf16_piece_busy_share"] or 0.0) * r["mfma_busy_cycles_per_step"] for r in conv) / busy, 4) if busy else None}
        if algo_gflop:
            summary["conv_family_algorithmic_gflop_per_step"] = algo_gflop
            summary["executed_over_algorithmic"] = round(ex / algo_gflop, 3)
        if all("effective_clock_ghz" in r for r in conv):
            ms = sum(r["kernel_ms_per_step_this_pass"] for r in conv)
            summary["conv_family_effective_clock_ghz"] = round(cyc / (ms * 1e6), 3)
            summary["conv_family_ms_per_step_this_pass"] = round(ms, 2)
        total_cyc = sum(r["kernel_cycles_per_step"] for r in res.values())
        summary["whole_step_mfma_busy_frac"] = round(sum(r["mfma_busy_cycles_per_step"] for r in res.values()) / (total_cyc * SIMDS), 4)
    for r in res.values():
        r.pop("bf16_pflops_while_running", None)
    json.dump({"normalisation": "MFMA busy cycles (sum over SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); executed FLOP in fp32-product equivalents = 64 x busy cycles of the "
                                "fp32-MFMA kernels (32x32x2: 64 cycles, 4096 FLOP) + 1024 / 6 x busy cycles of the bf16x3 kernels (16x16x32 bf16: 16 cycles, 16384 FLOP, six piece "
                                "products per fp32 product)",
               "steps_in_run": steps, **build_identity(), "summary": summary, "per_class": res}, open(out, "w"), indent=1)
    print(json.dumps({"summary": summary, "per_class": res}, indent=1))


if __name__ == "__main__":
    main()

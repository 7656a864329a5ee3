#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, --output-format csv) into HBM GB per training
step and kernel class.  Corrections as MI355X_MICROARCH.md prescribes: both counters are in KiB; FETCH_SIZE is doubled on
gfx950.  usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <steps in the run> <out.json>"""
import csv
import hashlib
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def build_identity():
    """Which build the profiled run loaded (SSV_HIP_LIB or the in-tree library): {"src_sha16": the sources it was compiled from, as the binary
    itself reports (ssv_source_sha16), "lib_sha16": sha256[:16] of the file}.  bench.py refuses counters whose src_sha16 is not its library's."""
    import ctypes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.environ.get("SSV_HIP_LIB") or os.path.join(root, "self-supervised-vision_amd", "csrc", "libssv_hip.so")
    with open(path, "rb") as fh:
        file_sha = hashlib.sha256(fh.read()).hexdigest()[:16]
    fn = ctypes.CDLL(path).ssv_source_sha16
    fn.restype = ctypes.c_char_p
    return {"src_sha16": fn().decode(), "lib_sha16": file_sha}


from kernel_classes import classify  # noqa: E402  (one table for every aggregator)


def total(path, counter):
    acc, n = defaultdict(float), defaultdict(int)
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            c = classify(row["Kernel_Name"])
            acc[c] += float(row["Counter_Value"])
            n[c] += 1
    return acc, n


def main():
    fetch_csv, write_csv, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, nf = total(fetch_csv, "FETCH_SIZE")
    w, _ = total(write_csv, "WRITE_SIZE")
    per = {c: {"fetch": round(2.0 * f[c] * 1024 / 1e9 / steps, 2), "write": round(w[c] * 1024 / 1e9 / steps, 2), "launches": nf[c] // steps}
           for c in sorted(set(f) | set(w))}
    json.dump({"corrections": "counters are KiB; FETCH_SIZE doubled (gfx950); divided by the number of steps in the profiled run",
               "steps_in_run": steps, **build_identity(), "per_step_gb": per}, open(out, "w"), indent=1)
    print(json.dumps(per, indent=1))


if __name__ == "__main__":
    main()

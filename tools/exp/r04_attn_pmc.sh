#!/bin/bash
# attention kernels alone (tools/bench_attn.py): matrix-pipe busy, wave stall buckets, LDS bank conflicts and the clock the chip held
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/bench_attn.py 10
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_attn -o run -- python3 tools/bench_attn.py 5 > gpurun_out/pmc_attn.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_attn/**/*counter_collection.csv", recursive=True)
t = glob.glob("gpurun_out/pmc_attn/**/*kernel_trace.csv", recursive=True)
disp = {}
for r in csv.DictReader(open(t[0])):
    disp[r["Dispatch_Id"]] = (r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3, r.get("Grid_Size", r.get("Grid_Size_X", "")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float)
seen = set()
for r in csv.DictReader(open(f[0])):
    name, us, grid = disp[r["Dispatch_Id"]]
    if "attn" not in name: continue
    import re
    k = (re.search(r"attn_\w+<\d+>", name).group(0), grid)
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); cnt[k] += 1; dur[k] += us
for k, a in sorted(agg.items()):
    n = cnt[k]; gui = a["GRBM_GUI_ACTIVE"] / 8 / n; us = dur[k] / n; wc = a["SQ_WAVE_CYCLES"]
    print("%-30s grid %-9s n=%d us=%.0f clk=%.2fGHz mfma_busy=%.3f wait_any=%.3f wait_inst=%.3f (lds %.3f) active=%.3f lds_conflict/wave_cyc=%.3f" % (
        k[0], k[1], n, us, gui / us / 1e3, a["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (gui * 1024), a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc,
        a["SQ_WAIT_INST_LDS"] / wc, a["SQ_ACTIVE_INST_ANY"] / wc, a["SQ_LDS_BANK_CONFLICT"] / wc))
PY

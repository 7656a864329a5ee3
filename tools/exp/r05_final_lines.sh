#!/bin/bash
# Evidence of round 5, part 3: the bench lines on the current library - the driver's default line (SimCLR + config3_rank_emulation + other_configs), --emulate-world 8,
# and the three other BASELINE workloads as their own lines.  LABEL prefixes the files (default r05_b)
L=${1:-r05_b}
python bench.py --steps 20 --warmup 5 > gpurun_out/${L}_bench_simclr.json 2> gpurun_out/${L}_bench_simclr.err
python bench.py --emulate-world 8 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${L}_bench_simclr_emulate_world8.json 2> gpurun_out/${L}_bench_simclr_emulate_world8.err
for a in dino byol barlow; do
  timeout -k 10 500 python bench.py --algo $a --steps 10 --warmup 3 > gpurun_out/${L}_bench_$a.json 2> gpurun_out/${L}_bench_$a.err
done
python3 - <<PY
import json
for a in ("simclr", "dino", "byol", "barlow"):
    d = json.load(open(f"gpurun_out/${L}_bench_{a}.json")); r = d["roofline"]; g = d["parity_gate"]
    print(a, d["value"], d["ms_per_step"], "frac", r["frac"], "executed", r.get("executed_frac"), "stale", r.get("counters_stale"), "traffic", r.get("traffic"), "whole", r["whole_step_mfma_frac"],
          "gate", g["pass"], g.get("loss_rel_err_teacher_forced", g.get("loss_rel_err")), g.get("update_at_config_lr"))
d = json.load(open("gpurun_out/${L}_bench_simclr.json"))
print("config3", {k: d["config3_rank_emulation"].get(k) for k in ("ms_per_step", "compute_side_scaling_ceiling")}, d["config3_rank_emulation"].get("ntxent"))
print("other", {k: (v.get("value"), v.get("pass"), v.get("leg_seconds")) for k, v in d["other_configs"].items()})
print("config1", d["config1"]["gpu"]["value"], d["config1"]["gpu"]["eager"]["value"], d["config1"]["cpu"]["value"], d["config1"]["loss_step0"])
PY

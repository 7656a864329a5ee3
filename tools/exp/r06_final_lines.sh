#!/bin/bash
# Evidence of round 6, part 3: the bench lines on the current library - the driver's default line, --emulate-world 8, and the three other BASELINE workloads as their own lines.
L=${1:-r06_b}
python bench.py --steps 20 --warmup 5 > gpurun_out/${L}_bench_simclr.json 2> gpurun_out/${L}_bench_simclr.err
python bench.py --emulate-world 8 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${L}_bench_simclr_emulate_world8.json 2> gpurun_out/${L}_bench_simclr_emulate_world8.err
for a in dino byol barlow; do
  timeout -k 10 500 python bench.py --algo $a --steps 10 --warmup 3 > gpurun_out/${L}_bench_$a.json 2> gpurun_out/${L}_bench_$a.err
done
python3 - <<PY
import json
for a in ("simclr", "dino", "byol", "barlow"):
    d = json.load(open(f"gpurun_out/${L}_bench_{a}.json")); r = d["roofline"]; g = d["parity_gate"]
    print(a, d["value"], d["ms_per_step"], "frac", r["frac"], "vs fp32 roof", r.get("frac_vs_fp32_mfma_roof"), "executed", r.get("executed_frac"), "stale", r.get("counters_stale"), "traffic", r.get("traffic"),
          "fp32 path", (d.get("fp32_mfma_instruction_path") or {}).get("value"), "gate", g["pass"], g.get("loss_rel_err_teacher_forced", g.get("loss_rel_err")))
d = json.load(open("gpurun_out/${L}_bench_simclr.json"))
print("config3", {k: d["config3_rank_emulation"].get(k) for k in ("ms_per_step", "compute_side_scaling_ceiling")})
print("other", {k: (v.get("value"), v.get("pass"), v.get("leg_seconds")) for k, v in d["other_configs"].items()})
print("config1", d["config1"]["gpu"]["value"], d["config1"]["gpu"]["ms_per_step"], d["config1"]["cpu"]["value"], d["config1"]["loss_step0"])
print("knn", {k: d["eval_knn"].get(k) for k in ("ms_per_call", "kernel_ms", "tflops", "frac", "speedup_vs_unfused", "count_difference_vs_unfused")}, d["eval_knn"].get("unfused_fp32_path"))
PY

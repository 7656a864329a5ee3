# the four bench lines of the BASELINE workloads on the final library of round 4 (each with cpu_baseline + parity_gate and the replayed counters of this build)
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_b_bench_simclr.json 2> gpurun_out/r04_b_bench_simclr.err
for a in dino byol barlow; do
  timeout -k 10 400 python bench.py --algo $a --steps 10 --warmup 3 > gpurun_out/r04_b_bench_$a.json 2> gpurun_out/r04_b_bench_$a.err
done
python3 - <<'PY'
import json
for a in ("simclr", "dino", "byol", "barlow"):
    d = json.load(open(f"gpurun_out/r04_b_bench_{a}.json")); r = d["roofline"]; g = d["parity_gate"]
    print(a, d["value"], d["ms_per_step"], "frac", r["frac"], "executed", r.get("executed_frac"), "stale", r.get("counters_stale"), "traffic", r.get("traffic"), "whole", r["whole_step_mfma_frac"],
          "gate", g["pass"], g.get("loss_rel_err_teacher_forced", g.get("loss_rel_err")))
PY

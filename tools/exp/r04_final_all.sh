# final evidence of round 4 on the final library: profiles of both bench workloads, the 53-layer table, the four bench lines (the full GPU suite ran green on this
# library right before: gpurun_out/r04_suite.log)
export TMPDIR=/tmp
bash tools/profile_step.sh r04_simclr_b512 > gpurun_out/r04_profile_simclr.log 2>&1
bash tools/profile_step.sh r04_dino_b128 --algo dino > gpurun_out/r04_profile_dino.log 2>&1
timeout -k 10 300 python tools/bench_conv.py 512 5 gpurun_out/r04_conv_layers_b512.csv > gpurun_out/r04_conv_layers_b512.txt 2>&1; tail -3 gpurun_out/r04_conv_layers_b512.txt
for t in r04_simclr_b512 r04_dino_b128; do for f in kernel_stats_single_stream.csv kernel_stats_two_streams.csv pmc_hbm_traffic.json pmc_mfma.json family_time_per_step.txt; do cp gpurun_out/${t}_$f profiles/; done; done; cp gpurun_out/r04_conv_layers_b512.csv profiles/
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_e_bench_simclr.json 2> gpurun_out/r04_e_bench_simclr.err
for a in dino byol barlow; do
  timeout -k 10 400 python bench.py --algo $a --steps 10 --warmup 3 > gpurun_out/r04_e_bench_$a.json 2> gpurun_out/r04_e_bench_$a.err
done
python3 - <<'PY'
import json
for a in ("simclr", "dino", "byol", "barlow"):
    d = json.load(open(f"gpurun_out/r04_e_bench_{a}.json")); r = d["roofline"]; g = d["parity_gate"]
    print(a, d["value"], d["ms_per_step"], "frac", r["frac"], "executed", r.get("executed_frac"), "stale", r.get("counters_stale"), "traffic", r.get("traffic"), "whole", r["whole_step_mfma_frac"],
          "gate", g["pass"], g.get("loss_rel_err_teacher_forced", g.get("loss_rel_err")), g.get("z_err_vs_fp64"))
PY

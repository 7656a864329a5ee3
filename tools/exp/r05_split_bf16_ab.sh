#!/bin/bash
# r05 e15: the Winograd forward / data-gradient products on the BF16 pipe by operand splitting (SSV_SPLIT_BF16=6, opt-in) against the shipped fp32-MFMA products:
# same-box alternating pairs of the headline step, then ONE full line with the switch on (parity gate, teacher-forced 1e-4, on the split products)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for i in 1 2 3 4 5; do
  for v in 0 6; do
    L=$(SSV_SPLIT_BF16=$v python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --prof-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
    echo "pair $i SSV_SPLIT_BF16=$v: ms_per_step images/s = $L"
  done
done
SSV_SPLIT_BF16=6 python3 bench.py --no-other-configs > gpurun_out/r05/split6_bench_simclr.json 2>/dev/null
python3 - <<PY
import json
d = json.load(open("gpurun_out/r05/split6_bench_simclr.json"))
g = d["parity_gate"]
print("full line with SSV_SPLIT_BF16=6:", d["value"], "images/s", d["ms_per_step"], "ms; teacher-forced loss errors", g["loss_rel_err_teacher_forced"], "pass", g["teacher_forced_pass"], "gate pass", g.get("pass"), "dispatch", g.get("dispatch"))
print("switches recorded:", d["config"]["diagnostic_switches"])
PY

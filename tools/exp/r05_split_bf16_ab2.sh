#!/bin/bash
# r05 e15b: SSV_SPLIT_BF16=6 widened to every plain 1x1 / Linear forward and data-gradient product (>= 128 output channels) besides the Winograd products: DINO ViT-S/16
# (GEMM-bound) and the SimCLR headline step, same-box alternating pairs; then one full DINO line with the switch on (its 3-step free-running parity gate on the split products)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for a in dino simclr; do
  for i in 1 2 3; do
    for v in 0 6; do
      L=$(SSV_SPLIT_BF16=$v python3 bench.py --algo $a --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --prof-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
      echo "$a pair $i SSV_SPLIT_BF16=$v: ms_per_step images/s = $L"
    done
  done
done
SSV_SPLIT_BF16=6 python3 bench.py --algo dino > gpurun_out/r05/split6_bench_dino.json 2>/dev/null
python3 - <<PY
import json
d = json.load(open("gpurun_out/r05/split6_bench_dino.json"))
g = d["parity_gate"]
print("full DINO line with SSV_SPLIT_BF16=6:", d["value"], "images/s", d["ms_per_step"], "ms; gate", g.get("loss_rel_err"), "pass", g.get("pass"))
PY

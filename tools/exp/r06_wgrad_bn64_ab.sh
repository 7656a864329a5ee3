B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-arith-legs --prof-steps 1"
for i in 1 2; do
for v in shipped wgbn64; do
  if [ $v = shipped ]; then $B > gpurun_out/_ab.json 2>/dev/null; else SSV_HIP_LIB=tools/probe/bin/libssv_wgbn64.so $B > gpurun_out/_ab.json 2>/dev/null; fi
  python - <<PY
import json
d=json.load(open("gpurun_out/_ab.json")); c=d["roofline"]["classes"]
print("$v", d["value"], d["ms_per_step"], "wgrad", c["conv_wgrad"]["ms_per_step"], "fwd", c["conv_fwd"]["ms_per_step"])
PY
done; done

# the four bench lines of the BASELINE workloads on the final library of round 4 (each with cpu_baseline + parity_gate), after the DINO profile
export TMPDIR=/tmp
bash tools/profile_step.sh r04_dino_b128 --algo dino > gpurun_out/r04_profile_dino.log 2>&1
tail -3 gpurun_out/r04_profile_dino.log
for a in dino byol barlow; do
  timeout -k 10 400 python bench.py --algo $a --steps 10 --warmup 3 > gpurun_out/r04_a_bench_$a.json 2> gpurun_out/r04_a_bench_$a.err
  python3 -c "import json; d=json.load(open('gpurun_out/r04_a_bench_$a.json')); print('$a', d['value'], d['ms_per_step'], d['parity_gate']['pass'] if d.get('parity_gate') else None)"
done

#!/bin/bash
# sample rocm-smi while the bench runs
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --prof-steps 0 > gpurun_out/smi_bench.log 2>&1 &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power|Temperature" | head -12; echo ---; sleep 1; done > gpurun_out/smi_samples.txt 2>&1
wait $BP
echo idle >> gpurun_out/smi_samples.txt
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power" | head -8 >> gpurun_out/smi_samples.txt
tail -1 gpurun_out/smi_bench.log | cut -c1-120

#!/bin/bash
# r06: what is left of config 1's graphed step (ResNet-18, 32x32, batch 64) on the bf16x3 arithmetic: kernel trace of graph replays grouped by (kernel, workgroups)
export TMPDIR=/tmp
OUT=gpurun_out
rm -rf $OUT/_prof_cg
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_prof_cg -- python3 tools/bench_cifar.py 64 graph_steps 20 > $OUT/r06_cifar_graph_b64.log 2>&1 || exit 1
tail -1 $OUT/r06_cifar_graph_b64.log
t=$(find $OUT/_prof_cg -name '*kernel_trace.csv' | head -1)
python3 tools/kstats_steady.py "$t" > $OUT/r06_cifar_graph_b64_family_time.txt
tail -3 $OUT/r06_cifar_graph_b64_family_time.txt
n=$(grep -o '[0-9]* launches' $OUT/r06_cifar_graph_b64_family_time.txt | tail -1 | cut -d' ' -f1)
python3 tools/trace_by_grid.py "$t" $n 3 > $OUT/r06_cifar_graph_b64_by_grid.txt
head -45 $OUT/r06_cifar_graph_b64_by_grid.txt | cut -c1-200

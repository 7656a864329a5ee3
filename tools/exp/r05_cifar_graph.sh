#!/bin/bash
# r05 e6: the CIFAR regime through the step graph: kernel trace of graph replays at bs 64 / 512 (where does the GPU time of a launch-free step go?)
set -u
OUT=gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
for bs in 64 512; do
  rm -rf $OUT/_prof_cg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_prof_cg -- python3 tools/bench_cifar.py $bs graph_steps 20 > $OUT/r05_cifar_graph_b${bs}.log 2>&1
  tail -1 $OUT/r05_cifar_graph_b${bs}.log
  f=$(find $OUT/_prof_cg -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/r05_cifar_graph_b${bs}_kernel_stats.csv
  t=$(find $OUT/_prof_cg -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python3 tools/kstats_steady.py "$t" > $OUT/r05_cifar_graph_b${bs}_family_time.txt
done
rm -rf $OUT/_prof_cg
head -25 $OUT/r05_cifar_graph_b64_kernel_stats.csv | cut -c1-230
tail -4 $OUT/r05_cifar_graph_b64_family_time.txt

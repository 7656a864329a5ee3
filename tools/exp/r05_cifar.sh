#!/bin/bash
# r05 e4: the reference's own shipped regime - SimCLR resnet18 (reduce_bottom_conv) on 32x32 images (configs/simclr.yaml) at bs 512 and bs 64:
# bench lines, rocprofv3 kernel trace (stats, per-step family time), one counter pass per variant table.  TAG prefixes the outputs (default r05_cifar_r18).
set -u
TAG=${1:-r05_cifar_r18}
OUT=gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
for bs in 512 64; do
  python tools/bench_cifar.py $bs > $OUT/${TAG}_b${bs}_bench.json 2> $OUT/${TAG}_b${bs}_bench.err || { tail -20 $OUT/${TAG}_b${bs}_bench.err; exit 1; }
  cat $OUT/${TAG}_b${bs}_bench.json
done
for bs in 512 64; do
  rm -rf $OUT/_prof_cifar
  SSV_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_prof_cifar -- python3 tools/bench_cifar.py $bs steps 4 > $OUT/_prof_cifar.log 2>&1
  f=$(find $OUT/_prof_cifar -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/${TAG}_b${bs}_kernel_stats.csv
  t=$(find $OUT/_prof_cifar -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python3 tools/kstats_steady.py "$t" > $OUT/${TAG}_b${bs}_family_time.txt
  rm -rf $OUT/_pmc_cifar
  SSV_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/_pmc_cifar -- python3 tools/bench_cifar.py $bs steps 2 > $OUT/_pmc_cifar.log 2>&1
  c=$(find $OUT/_pmc_cifar -name '*counter_collection.csv' | head -1); t=$(find $OUT/_pmc_cifar -name '*kernel_trace.csv' | head -1)
  [ -n "$c" ] && [ -n "$t" ] && python3 tools/pmc_variants.py "$c" "$t" 5 45 > $OUT/${TAG}_b${bs}_kernel_variants.txt
  [ -n "$c" ] && (cd tools && python3 pmc_mfma.py "../$c" 5 ../$OUT/${TAG}_b${bs}_pmc_mfma.json - "../$t" > /dev/null)
done
rm -rf $OUT/_prof_cifar $OUT/_pmc_cifar
ls -la $OUT/${TAG}_*

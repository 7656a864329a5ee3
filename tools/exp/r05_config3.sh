#!/bin/bash
# r05 e1: config 3's per-rank shape - parity tests, then the default bench line (config3_rank_emulation + other_configs) and --emulate-world 8
set -e
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_config3.py tests/test_gpu_ops.py -m gpu -x -q -k "config3 or ntxent or rank or replicated or splits" > gpurun_out/r05/e1_tests.log 2>&1 || { tail -40 gpurun_out/r05/e1_tests.log; exit 1; }
tail -3 gpurun_out/r05/e1_tests.log
T0=$SECONDS
python bench.py > gpurun_out/r05/e1_bench_default.json 2> gpurun_out/r05/e1_bench_default.err || { tail -30 gpurun_out/r05/e1_bench_default.err; exit 1; }
echo "default bench.py wall seconds: $((SECONDS - T0))" | tee gpurun_out/r05/e1_bench_default.wall
python bench.py --emulate-world 8 --no-cpu-baseline > gpurun_out/r05/e1_bench_emulate8.json 2> gpurun_out/r05/e1_bench_emulate8.err || { tail -30 gpurun_out/r05/e1_bench_emulate8.err; exit 1; }
SSV_NTXENT_SPLITS=1 python bench.py --emulate-world 8 --no-cpu-baseline --prof-steps 0 > gpurun_out/r05/e1_bench_emulate8_unsplit.json 2> gpurun_out/r05/e1_bench_emulate8_unsplit.err || { tail -30 gpurun_out/r05/e1_bench_emulate8_unsplit.err; exit 1; }
echo done

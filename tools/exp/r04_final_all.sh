# final evidence of round 4 on the final library: full GPU suite, smoke, profiles of both bench workloads, the 53-layer table, the four bench lines
export TMPDIR=/tmp
python -m pytest tests -q -m gpu > gpurun_out/r04_suite.log 2>&1; tail -3 gpurun_out/r04_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_step.sh r04_simclr_b512 > gpurun_out/r04_profile_simclr.log 2>&1
bash tools/profile_step.sh r04_dino_b128 --algo dino > gpurun_out/r04_profile_dino.log 2>&1
timeout -k 10 300 python tools/bench_conv.py 512 5 gpurun_out/r04_conv_layers_b512.csv > gpurun_out/r04_conv_layers_b512.txt 2>&1; tail -3 gpurun_out/r04_conv_layers_b512.txt
ls gpurun_out/r04_*pmc* gpurun_out/r04_*family*

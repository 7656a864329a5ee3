set -u
S2=tools/probe/bin/libssv_s2k12.so
echo "== correctness of the S2 side library (ViT ops + DINO step tests)"
SSV_HIP_LIB=$S2 timeout -k 10 600 python -m pytest tests/test_gpu_vit_ops.py tests/test_gpu_dino.py -x -q -m gpu 2>&1 | tail -3
echo "== GEMMs alone: shipped"
timeout -k 10 200 python tools/bench_vit_gemm.py 10 2>&1 | tail -9
echo "== GEMMs alone: S2 (>= 12 k-tiles)"
SSV_HIP_LIB=$S2 timeout -k 10 200 python tools/bench_vit_gemm.py 10 2>&1 | tail -9
for i in 1 2; do
  for v in shipped s2; do
    if [ $v = s2 ]; then export SSV_HIP_LIB=$S2; else unset SSV_HIP_LIB; fi
    timeout -k 10 300 python bench.py --algo dino --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dino $v $i', d['value'], d['ms_per_step'])"
  done
done
unset SSV_HIP_LIB

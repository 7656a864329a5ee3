# GELU epilogues of the ViT FFN: the library's own erf (shipped) vs libm erff (side library -DSSV_LIBM_ERF=1; note: only conv_mfma / bn are rebuilt in a side library, the
# stand-alone GELU kernels keep the shipped erf), and the derivative taken in the forward (shipped) vs in the backward epilogue (SSV_NO_GELU_DACT=1)
echo "== correctness"
timeout -k 10 600 python -m pytest tests/test_gpu_vit_ops.py tests/test_gpu_dino.py -x -q -m gpu 2>&1 | tail -3
echo "== epilogue probe: shipped erf"
timeout -k 10 120 python tools/probe/gelu_epilogue_probe.py 2>&1 | grep -v amdgpu.ids
echo "== epilogue probe: libm erff"
SSV_HIP_LIB=tools/probe/bin/libssv_libmerf.so timeout -k 10 120 python tools/probe/gelu_epilogue_probe.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
  for v in shipped nodact libm_nodact; do
    unset SSV_HIP_LIB SSV_NO_GELU_DACT
    [ $v = nodact ] && export SSV_NO_GELU_DACT=1
    [ $v = libm_nodact ] && export SSV_NO_GELU_DACT=1 SSV_HIP_LIB=tools/probe/bin/libssv_libmerf.so
    timeout -k 10 300 python bench.py --algo dino --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dino $v $i', d['value'], d['ms_per_step'])"
  done
done

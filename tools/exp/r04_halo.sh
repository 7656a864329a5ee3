#!/bin/bash
# The halo loader for the 64-channel 3x3 layers (conv_fwd_k, C4 == 3; diagnostic builds only) against the shipped generic loader.  Side libraries first:
#   tools/probe/build_variant.sh halo        -DSSV_EXP_HALO
#   tools/probe/build_variant.sh halow1      -DSSV_EXP_HALO -DSSV_WHATIF=1      (a quarter of the MFMAs)     halow8: -DSSV_WHATIF=8 (filter loads from one address)
#   tools/probe/build_variant.sh stg_halo12  -DSSV_EXP_HALO -DSSV_EXP_STAGGER=12 -DSSV_EXP_STAGGER_N=2      stg12: -DSSV_EXP_STAGGER=12 (generic loader)
# 1. unit tests on both, 2. the layer's products, 3. counters (matrix-pipe busy, wave stall buckets, effective clock = GRBM_GUI_ACTIVE / 8 / kernel time)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_halo.py -x -q -m gpu 2>&1 | tail -2
SSV_HIP_LIB=tools/probe/bin/libssv_halo.so timeout -k 10 300 python -m pytest tests/test_gpu_halo.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -2
for v in shipped halo ${SSV_HALO_VARIANTS}; do
  if [ $v = shipped ]; then unset SSV_HIP_LIB; else export SSV_HIP_LIB=tools/probe/bin/libssv_$v.so; fi
  SSV_BENCH_LAYERS=p64.0.conv2 python tools/bench_conv.py 512 10 gpurun_out/r04_halo_layer.csv > /dev/null 2>&1
  echo "variant=$v $(sed -n 2p gpurun_out/r04_halo_layer.csv | cut -d, -f1,9-11,14-16)"
done
for v in shipped halo; do
  if [ $v = shipped ]; then unset SSV_HIP_LIB; else export SSV_HIP_LIB=tools/probe/bin/libssv_$v.so; fi
  SSV_BENCH_LAYERS=p64.0.conv2 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_$v -o run -- python3 tools/bench_conv.py 512 10 > gpurun_out/pmc_$v.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for v in ("shipped", "halo"):
    f = glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % v, recursive=True)
    t = glob.glob("gpurun_out/pmc_%s/**/*kernel_trace.csv" % v, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"][:90]][r["Counter_Name"]] += float(r["Counter_Value"])
    for r in csv.DictReader(open(t[0])):
        k = r["Kernel_Name"][:90]; dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3; cnt[k] += 1
    for k, a in agg.items():
        if "conv_" not in k: continue
        n = cnt[k]; gui = a["GRBM_GUI_ACTIVE"] / 8 / n; us = dur[k] / n
        print(v, k[40:90], "n=%d us=%.0f clk=%.2fGHz mfma_busy=%.3f wait_any=%.3f wait_inst=%.3f active=%.3f" % (
            n, us, gui / us / 1e3, a["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (gui * 1024),
            a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"], a["SQ_ACTIVE_INST_ANY"] / a["SQ_WAVE_CYCLES"]))
PY

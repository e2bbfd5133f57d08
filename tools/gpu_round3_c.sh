#!/bin/bash
# round 3, session C: dependency-cost probes (flags vs launches, LDS-resident arithmetic decoder), the split
# (partially batched) transforms, 1x1 kernel variants
set -e
mkdir -p gpurun_out
timeout -k 10 120 tools/_build/flag_chain_probe 780 > gpurun_out/r3c_flag_chain.txt 2>&1 || { cat gpurun_out/r3c_flag_chain.txt; echo "flag probe failed"; }
cat gpurun_out/r3c_flag_chain.txt
timeout -k 10 120 tools/_build/ac_device_probe > gpurun_out/r3c_ac_probe.txt 2>&1 || { echo "ac probe failed"; }
cat gpurun_out/r3c_ac_probe.txt
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_codec_vs_oracle.py -m gpu -x -q -k "not metric_size" > gpurun_out/r3c_pytest.log 2>&1 || { tail -40 gpurun_out/r3c_pytest.log; exit 1; }
tail -3 gpurun_out/r3c_pytest.log
OUT=gpurun_out/r3c_split.txt
: > $OUT
for rep in 1 2; do
for cfg in "0 0" "6 5" "3 8" "6 8" "3 5"; do
  set -- $cfg
  echo "== rep $rep analysis_split $1 synthesis_split $2" >> $OUT
  PCONV_ANALYSIS_SPLIT=$1 PCONV_SYNTHESIS_SPLIT=$2 timeout -k 10 300 python bench.py --steps 3 --no-cpu-baseline 2>>gpurun_out/r3c_err.log \
    | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'MPix/s', d['ms_per_step'], 'ms/step', 'conv_s', d['config']['tile_conv_s_per_step'])" >> $OUT
  tail -1 $OUT
done
done
for v in base W2R8 W5R8 W6R8; do
  echo "== 1x1 variant $v" >> gpurun_out/r3c_1x1.txt
  if [ $v = base ]; then unset PCONV_HIP_LIB; else export PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$v.so; fi
  timeout -k 10 120 python tools/gpu_probe_1x1.py 2>/dev/null >> gpurun_out/r3c_1x1.txt
done
unset PCONV_HIP_LIB
cat gpurun_out/r3c_1x1.txt
echo done

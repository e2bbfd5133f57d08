#!/bin/bash
export PROBE_TN=128
for rep in 1 2; do for v in base qa2 qa6 qa8; do
  if [ $v = base ]; then unset PCONV_HIP_LIB; else export PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$v.so; fi
  echo "== $v rep $rep"; timeout -k 10 120 python tools/gpu_probe_1x1.py 2>&1 | grep -v "Warning\|amdgpu.ids"
done; done 2>&1 | tee gpurun_out/r4ag_quad_ahead.txt

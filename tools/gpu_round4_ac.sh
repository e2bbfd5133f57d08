#!/bin/bash
for rep in 1 2; do for v in base rc8 rc32 rc64; do
  if [ $v = base ]; then unset PCONV_HIP_LIB; else export PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$v.so; fi
  echo "== $v rep $rep"; timeout -k 10 120 python tools/gpu_probe_hbm.py 2>&1 | grep "ring"
done; done 2>&1 | tee gpurun_out/r4ac_ring_channels.txt

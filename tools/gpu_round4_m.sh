#!/bin/bash
for rep in 1 2; do for v in w42a w42s w42ns; do
  echo "== $v rep $rep"; PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$v.so PCONV_PROBE_SHORT=1 PCONV_PROBE_NODIRECT=1 timeout -k 10 120 python tools/gpu_probe_wino42.py 2>&1 | grep "3x3" | sed 's/wino [0-9.]* ms ([0-9]* TF alg, diff 0)//'
done; done 2>&1 | tee gpurun_out/r4m_stagger.txt

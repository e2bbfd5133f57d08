#!/bin/bash
# 1x1 / GDN classes: K-chunk size and operand prefetch depth of the tiled kernel (variant libraries)
for rep in 1 2; do for v in base kc32 kc8 pf2; do
  if [ $v = base ]; then unset PCONV_HIP_LIB; else export PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$v.so; fi
  echo "== $v rep $rep"; timeout -k 10 120 python tools/gpu_probe_1x1.py 2>&1 | grep -v Warning
done; done 2>&1 | tee gpurun_out/r4r_1x1_chunks.txt

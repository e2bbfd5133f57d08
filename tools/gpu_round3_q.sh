#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/r3q_wino.txt
for v in base WFP base WFP; do
  echo "== variant $v" >> gpurun_out/r3q_wino.txt
  if [ $v = base ]; then unset PCONV_HIP_LIB; else export PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$v.so; fi
  timeout -k 10 300 python tools/gpu_probe_wino.py 2>/dev/null | cut -c1-125 | sed -n '1p;5p;7p' >> gpurun_out/r3q_wino.txt
done
cat gpurun_out/r3q_wino.txt

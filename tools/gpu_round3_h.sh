#!/bin/bash
# round 3, session H: Winograd main-loop ablations (timing only: wrong results)
mkdir -p gpurun_out
: > gpurun_out/r3h_wino.txt
for v in WD0 WD1 WD2 WD3 WD4; do
  echo "== variant $v" >> gpurun_out/r3h_wino.txt
  export PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$v.so
  timeout -k 10 300 python tools/gpu_probe_wino.py 2>/dev/null | cut -c1-118 | sed -n '1p;5p;7p' >> gpurun_out/r3h_wino.txt
done
cat gpurun_out/r3h_wino.txt

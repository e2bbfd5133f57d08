#!/bin/bash
# time the 3x3 192->192 layer (half-resolution scale of a 4096x2048 frame) with each experiment build
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/ablate_conv.log
: > $out
for v in base "$@"; do
  echo "== $v" >> $out
  if [ $v = base ]; then unset PCONV_HIP_LIB; else export PCONV_HIP_LIB=$PWD/tools/_build/libpconv_hip_$v.so; fi
  timeout -k 10 120 python tools/gpu_probe_conv.py >> $out 2>&1 || { echo "FAILED $v" >> $out; break; }
done
cat $out | grep -v amdgpu.ids
